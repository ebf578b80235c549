#!/bin/bash
# round 6: GPT-3B tile GEMMs at 512 rows without the statistics prologue (mask 16) and skeleton without it (22 = 16 + 4 + 2); prefetch upper bound
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( ROWS=64 timeout 300 python tools/prefetch_bound.py 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee gpurun_out/r6_prefetch_bound.log
( ROWS=128 timeout 300 python tools/prefetch_bound.py 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee -a gpurun_out/r6_prefetch_bound.log
for mask in 16 22 30; do
  echo "== LGEN_TILE_ABLATE=$mask"
  LGEN_TILE_ABLATE=$mask LGEN_SWEEP_KINDS=qkv,w13 timeout 600 python tools/gemm_tile_sweep.py GPT-3B 512 2>&1 | grep "tile" | grep -v "rc " | awk '{ $NF=""; print }' | cut -c1-100
done 2>&1 | tee gpurun_out/r6_ablate3b2.log
