#!/bin/bash
# round 6: configs 3 (GPT-XXL, 2 x 512 rows) and 5 (GPT-XL t2i, 2 x 384 rows): whole-bench A/Bs of tile shapes on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule"
run() { echo -n "$1: "; local c=$2 st=$3; shift 3; env "$@" timeout 600 python bench.py $F --config $c --steps $st --warmup 4 2>gpurun_out/r6_c35ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r6_c35ab_err.log; }
{
run c3_base 3 16 X=1
run c3_qkv_4x2x6 3 16 "LGEN_TILE_SHAPES=qkv=4,1,2,6,2,4,4"
run c3_qkv_4x1x8 3 16 "LGEN_TILE_SHAPES=qkv=4,1,1,8,2,4,4"
run c3_w13_8x1x8 3 16 "LGEN_TILE_SHAPES=w13=8,1,1,8,2,4,4"
run c3_w13_4x2x6 3 16 "LGEN_TILE_SHAPES=w13=4,1,2,6,2,4,4"
run c3_wo_w2_2241 3 16 "LGEN_TILE_SHAPES=wo=2,2,4,1,4,4,4;w2=2,2,4,1,4,4,4"
run c3_wo_w2_2221 3 16 "LGEN_TILE_SHAPES=wo=2,2,2,1,4,4,4;w2=2,2,2,1,4,4,4"
run c3_base 3 16 X=1
run c5_base 5 24 X=1
run c5_qkv_4x1x6 5 24 "LGEN_TILE_SHAPES=qkv=4,1,1,6,4,3,4"
run c5_w13_4x2x8 5 24 "LGEN_TILE_SHAPES=w13=4,1,2,8,2,4,4"
run c5_w13_8x1x8 5 24 "LGEN_TILE_SHAPES=w13=8,1,1,8,2,4,4"
run c5_wo_w2_2222 5 24 "LGEN_TILE_SHAPES=wo=2,2,2,2,4,4,4;w2=2,2,2,2,4,4,4"
run c5_base 5 24 X=1
} 2>&1 | tee gpurun_out/r6_c35ab.log
