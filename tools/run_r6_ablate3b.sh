#!/bin/bash
# round 6: where does a launch's time go at GPT-3B widths (K = 3200 / 8704, 512 rows)?  LGEN_TILE_ABLATE masks: 1 no RMSNorm VALU, 2 no LDS reads / MFMAs,
# 4 no operand DMA, 6 = skeleton (barriers + prologue + epilogue only).  Then the prefetch upper bound for the literal batch-32 call.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( ROWS=64 timeout 300 python tools/prefetch_bound.py 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee gpurun_out/r6_prefetch_bound.log
( ROWS=128 timeout 300 python tools/prefetch_bound.py 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee -a gpurun_out/r6_prefetch_bound.log
for mask in 0 1 2 4 6; do
  echo "== LGEN_TILE_ABLATE=$mask"
  LGEN_TILE_ABLATE=$mask LGEN_SWEEP_KINDS=qkv,w13,w2 timeout 600 python tools/gemm_tile_sweep.py GPT-3B 512 2>&1 | grep "tile\|skinny" | grep -v "rc " | awk '{ $NF=""; print }' | cut -c1-100
done 2>&1 | tee gpurun_out/r6_ablate3b.log
