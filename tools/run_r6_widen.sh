#!/bin/bash
# round 6: do wider chains pay for configs 3 / 4 / 5 (and three chains of 7 for config 2)?  + tile sweeps at the new widths
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule"
run() { echo -n "$1: "; shift; env $ENVX timeout 600 python bench.py $F "$@" 2>gpurun_out/r6_widen_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['gemm_schedule']['schedule'])" || tail -3 gpurun_out/r6_widen_err.log; }
{
ENVX="X=1" run c3_2x6 --config 3 --steps 12
ENVX="X=1" run c3_2x8 --config 3 --steps 16 --batches-per-chain 8 --lanes 2
ENVX="X=1" run c3_2x10 --config 3 --steps 20 --batches-per-chain 10 --lanes 2
ENVX="X=1" run c4_2x2 --config 4 --steps 8
ENVX="X=1" run c4_2x4 --config 4 --steps 8 --batches-per-chain 4 --lanes 2
ENVX="X=1" run c5_3x4 --config 5 --steps 12
ENVX="X=1" run c5_2x8 --config 5 --steps 16 --batches-per-chain 8 --lanes 2
ENVX="X=1" run c5_2x6 --config 5 --steps 12 --batches-per-chain 6 --lanes 2
ENVX="X=1" run c2_2x10 --steps 20
ENVX="LGEN_TILE_SHAPES=qkv=4,1,1,8,2,4,4;wo=2,2,2,2,4,4,4;w13=4,1,1,8,2,4,4;w2=2,2,2,2,4,4,4;head=4,1,1,8,2,4,4" run c2_3x7 --steps 20 --batches-per-chain 7 --lanes 3
} 2>&1 | tee gpurun_out/r6_widen.log
for spec in "GPT-L 448" "GPT-XXL 512 640" "GPT-3B 512" "GPT-XL 256 192"; do
  timeout 600 python tools/gemm_tile_sweep.py $spec > gpurun_out/r6_sweep_$(echo $spec | tr ' ' '_').log 2>&1
done
grep -h "MISMATCH\|rc " gpurun_out/r6_sweep_*.log | head
