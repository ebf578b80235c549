#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_headline.py -q -x -k "tile_gemm_family or wide_chain or config4" 2>&1 | tail -5 ) | tee gpurun_out/r6_c4b_tests.log
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs"
run() { echo -n "$1: "; shift; timeout 900 python bench.py $F "$@" 2>gpurun_out/r6_c4b_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('roofline_gemm',{}); r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], d['config']['batches_per_chain'], d['config']['chains_in_flight_per_gpu'], 'attn', r.get('frac'), r.get('avg_launch_us'), 'gemm_us', g.get('us_per_step'), {k:v['us'] for k,v in g.get('per_launch',{}).items()}, 'launches', g.get('launches_per_step'))" || tail -3 gpurun_out/r6_c4b_err.log; }
{
run c4 --config 4 --steps 8 --warmup 2
run c4 --config 4 --steps 8 --warmup 2
} 2>&1 | tee gpurun_out/r6_c4b.log
