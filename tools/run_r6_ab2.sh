#!/bin/bash
# round 6 A/B: decode attention with packed dot products (v_dot2c_f32_bf16) for q.k  [liblgen_hip.so] against the previous build [liblgen_hip_base.so]
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_gpt.py -q -x -k "test_qkv_rope_append_and_attention" 2>&1 | tail -4 ) | tee gpurun_out/r6_ab2_tests.log
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs"
run() {
  echo -n "$1: "
  python tools/ab_lib.py llamagen_amd/$2 bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline',{}); g=d.get('roofline_gemm',{}); v=d.get('roofline_vq_decode',{})
print(d['value'], d['ms_per_step'], 'attn_frac', r.get('frac'), 'gemm_us', g.get('us_per_step'), 'vq_ms', v.get('ms_per_decode_code'))"
}
{
run base liblgen_hip_base.so
run dot2 liblgen_hip.so
run base liblgen_hip_base.so
run dot2 liblgen_hip.so
run base liblgen_hip_base.so
run dot2 liblgen_hip.so
} 2>&1 | tee gpurun_out/r6_ab2.log
( NO_V=1 ROWS=640 timeout 300 python tools/ab_lib.py llamagen_amd/liblgen_hip_base.so tools/overlap_probe.py 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r6_overlap_base.log 2>&1
( NO_V=1 ROWS=640 timeout 300 python tools/overlap_probe.py 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r6_overlap_dot2.log 2>&1
cat gpurun_out/r6_overlap_base.log gpurun_out/r6_overlap_dot2.log
