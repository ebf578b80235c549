#!/bin/bash
# round 6: pipelined fragment reads in the 3x3 fused conv (lgen_debug variant 2 / 3) against the default (1): one conv shape bit for bit + timed,
# whole decode_code(), the VQ parity tests under each variant, then the two-chain bench
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
{
timeout 300 python tools/conv_once.py 32 384 128 128 5 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/conv_once.py 32 192 256 256 5 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/conv_once.py 32 24 512 512 20 2>&1 | grep -v amdgpu.ids
for v in 1 2 3 1 2 3; do echo -n "LGEN_CF_VARIANT=$v "; LGEN_CF_VARIANT=$v timeout 300 python tools/vq_once.py 32 5 2>&1 | grep decode_code; done
for v in 2 3; do echo "== VQ tests LGEN_CF_VARIANT=$v"; LGEN_CF_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_vq.py tests/test_gpu_driver.py -q -x 2>&1 | tail -3; LGEN_CF_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_headline.py -q -x -k decode_code 2>&1 | tail -2; done
} 2>&1 | tee gpurun_out/r6_convpipe.log
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs"
run() { echo -n "$1: "; shift; env "$@" timeout 600 python bench.py $F 2>gpurun_out/r6_convpipe_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d.get('roofline_vq_decode',{}); print(d['value'], d['ms_per_step'], 'vq_ms', v.get('ms_per_decode_code'))" || tail -3 gpurun_out/r6_convpipe_err.log; }
{
run v1 LGEN_CF_VARIANT=1
run v2 LGEN_CF_VARIANT=2
run v3 LGEN_CF_VARIANT=3
run v1 LGEN_CF_VARIANT=1
run v2 LGEN_CF_VARIANT=2
run v3 LGEN_CF_VARIANT=3
} 2>&1 | tee gpurun_out/r6_convpipe_bench.log
