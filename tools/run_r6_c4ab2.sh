#!/bin/bash
# round 6: config 4 (GPT-3B, 2 x 512 rows), second whole-bench A/B on one box: the other GEMM kinds
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule --config 4 --steps 8 --warmup 2"
run() { echo -n "$1: "; shift; env "LGEN_TILE_SHAPES=$1" timeout 900 python bench.py $F 2>gpurun_out/r6_c4ab2_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r6_c4ab2_err.log; }
{
run base ""
run w13_8x1x6 "w13=8,1,1,6,2,4,4"
run w13_4x1x8 "w13=4,1,1,8,2,4,4"
run wo_w2_22222 "wo=2,2,2,2,2,4,4;w2=2,2,2,2,2,4,4"
run wo_w2_2241 "wo=2,2,4,1,4,4,4;w2=2,2,4,1,4,4,4"
run head_8x1x8 "head=8,1,1,8,2,4,4"
run qkv_4x2x6 "qkv=4,1,2,6,2,4,4"
run base ""
} 2>&1 | tee gpurun_out/r6_c4ab2.log
