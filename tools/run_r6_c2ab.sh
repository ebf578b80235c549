#!/bin/bash
# round 6: config 2 whole-bench A/Bs on one box: s_setprio in the tile GEMM (all waves 2 / loader waves 3), decode-attention variant 14 (98 VGPRs,
# 3 loads in flight), alternative tile shapes for w1||w3 / wqkv / wo / w2
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule"
run() { echo -n "$1: "; local lib=$2; shift 2; env "$@" timeout 600 python tools/ab_lib.py llamagen_amd/$lib bench.py $F 2>gpurun_out/r6_c2ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r6_c2ab_err.log; }
{
run base liblgen_hip.so X=1
run prio_all2 liblgen_hip_prio_all_2.so X=1
run prio_loader3 liblgen_hip_prio_loader_3.so X=1
run attn14 liblgen_hip.so LGEN_ATTN_VARIANT=14
run base liblgen_hip.so X=1
run w13_4x2x8 liblgen_hip.so "LGEN_TILE_SHAPES=w13=4,1,2,8,2,4,4"
run w13_4x1x8 liblgen_hip.so "LGEN_TILE_SHAPES=w13=4,1,1,8,2,4,4"
run qkv_8x1x6 liblgen_hip.so "LGEN_TILE_SHAPES=qkv=8,1,1,6,2,4,4"
run wo_w2_2242 liblgen_hip.so "LGEN_TILE_SHAPES=wo=2,2,4,2,2,4,4;w2=2,2,4,2,2,4,4"
run wo_w2_2241 liblgen_hip.so "LGEN_TILE_SHAPES=wo=2,2,4,1,4,4,4;w2=2,2,4,1,4,4,4"
run prio_all2 liblgen_hip_prio_all_2.so X=1
run base liblgen_hip.so X=1
} 2>&1 | tee gpurun_out/r6_c2ab.log
