cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['images_per_s_with_one_chain_in_flight'], d['images_per_s_with_one_step_in_flight']); print({k:(v.get('value'), v.get('error')) for k,v in d['other_configs'].items()}); print(d['roofline_gemm']['us_per_step'], d['roofline_vq_decode']['ms_per_decode_code'], d['cpu_baseline']['value'])"
