R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03_gpu_tests.log 2>&1; tail -5 gpurun_out/r03_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03_bench_line.json').read().strip().split('\n')[-1])
print({k: j.get(k) for k in ('value','ms_per_step','steps','images_per_s_with_one_chain_in_flight','images_per_s_with_one_step_in_flight')})
print('attn', j['roofline']['frac'], j['roofline']['avg_launch_us'], 'gemm', j['roofline_gemm']['frac'], j['roofline_gemm']['us_per_step'], 'vq', j['roofline_vq_decode']['ms_per_decode_code'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['thread_sweep_ms_per_step_min_of_3'], j['cpu_baseline']['c1']['seconds'])
PY
tail -2 gpurun_out/r03_bench_line.err
