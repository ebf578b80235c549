R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "passes_bit_identical" > gpurun_out/r3f_tests.log 2>&1; tail -3 gpurun_out/r3f_tests.log
timeout 600 python tools/smoke_configs.py > gpurun_out/r03_configs_3_4_5.log 2>&1; cat gpurun_out/r03_configs_3_4_5.log | grep -v amdgpu.ids
for c in 3 4 5; do
  timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-solo --no-live-traffic > gpurun_out/bench_config$c.json 2> gpurun_out/bench_config$c.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/bench_config$c.json').read().strip().split('\n')[-1])
    print('config $c:', j['value'], 'img/s; one chain', j['images_per_s_with_one_chain_in_flight'], '; attn frac', j['roofline']['frac'], '; gemm', j['roofline_gemm']['frac'], j['roofline_gemm']['us_per_step'], '; vq', j['roofline_vq_decode']['ms_per_decode_code'])
except Exception as e:
    print('config $c failed', e)
PY
  tail -2 gpurun_out/bench_config$c.err
done
bash tools/run_slow_parity.sh
