R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_vq.py tests/test_gpu_headline.py -x -q -k "not config and not qkv_fused" > gpurun_out/r3e_tests.log 2>&1
tail -12 gpurun_out/r3e_tests.log
timeout 300 python tools/vq_once.py 32 3 > gpurun_out/vq_once.log 2>&1; tail -5 gpurun_out/vq_once.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r3e.json 2> gpurun_out/bench_r3e.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_r3e.json').read().strip().split('\n')[-1])
print({k: j.get(k) for k in ('value','ms_per_step','images_per_s_with_one_chain_in_flight','images_per_s_with_one_step_in_flight')})
print('gemm', j['roofline_gemm']['us_per_step'], {k:v['us'] for k,v in j['roofline_gemm']['per_launch'].items()})
print('vq', j['roofline_vq_decode']['ms_per_decode_code'], 'attn', j['roofline']['frac'])
PY
tail -3 gpurun_out/bench_r3e.err
