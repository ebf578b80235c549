#!/bin/bash
# round 6: does the VQ decoder overlap better with a decode-attention variant that fits beside two conv waves per SIMD?
# (conv_fused 194 VGPRs x 2 per SIMD leave 112; variant 10 = 114 VGPRs, 14 = 98, 13 = 60 with 8 waves)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for v in 10 14 13 12; do
  echo "== ATTN_VARIANT=$v"
  ATTN_VARIANT=$v ROWS=640 timeout 400 python tools/overlap_probe.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | tail -8
done 2>&1 | tee gpurun_out/r6_va.log
