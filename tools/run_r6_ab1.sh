#!/bin/bash
# round 6 A/B: library without packed-fp32 instructions (-packed-fp32-ops) and / or s_setprio 3 in the persistent attention kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
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
run base liblgen_hip.so
run nopk liblgen_hip_nopk.so
run prio liblgen_hip_prio.so
run nopk_prio liblgen_hip_nopk_prio.so
run base liblgen_hip.so
run nopk liblgen_hip_nopk.so
run prio liblgen_hip_prio.so
run nopk_prio liblgen_hip_nopk_prio.so
} 2>&1 | tee gpurun_out/r6_ab1.log
