R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "pipeline" > gpurun_out/r3l_tests.log 2>&1; tail -3 gpurun_out/r3l_tests.log
for c in 3 4 5; do
  steps=$([ $c = 4 ] && echo 4 || echo 12)
  timeout 900 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-live-traffic > gpurun_out/bench_config$c.json 2> gpurun_out/bench_config$c.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/bench_config$c.json').read().strip().split('\n')[-1])
    print('config $c:', j['value'], 'img/s; chains', j['config']['chains_in_flight_per_gpu'], 'x', j['config']['batches_per_chain'], '; one chain', j['images_per_s_with_one_chain_in_flight'], '; one step', j.get('images_per_s_with_one_step_in_flight'), '; attn', j['roofline']['frac'], '; gemm', j['roofline_gemm']['frac'], j['roofline_gemm']['us_per_step'])
except Exception as e:
    print('config $c failed', e)
PY
  tail -2 gpurun_out/bench_config$c.err
done
