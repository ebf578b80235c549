#!/bin/bash
# round 6: config 4 (GPT-3B, 2 x 512 rows) same-box A/B of the wqkv / w1||w3 tile shapes: 128 x 160 (one round of workgroups) against 128 x 96
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule --config 4 --steps 8 --warmup 2"
run() { echo -n "$1: "; shift; env "LGEN_TILE_SHAPES=$1" timeout 900 python bench.py $F 2>gpurun_out/r6_c4ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r6_c4ab_err.log; }
OLD="qkv=8,1,1,6,2,4,4;w13=4,1,2,6,2,4,4"
QONLY="w13=4,1,2,6,2,4,4"
WONLY="qkv=8,1,1,6,2,4,4"
{
run new ""
run old "$OLD"
run new_qkv_only "$QONLY"
run new_w13_only "$WONLY"
run new ""
run old "$OLD"
} 2>&1 | tee gpurun_out/r6_c4ab.log
