# round 5, last evidence: (1) the bf16 free-running agreement report (oracle side precomputed), (2) rocprofv3 kernel stats of the bench
# with ONE 640-row chain in flight (the stand-alone duration of the dominant kernel at the timed chain shape), (3) smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_headline.py -q -x -k "free_running_agreement" 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r5_report_test.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_1x640 -- python $R/bench.py --lanes 1 --batches-per-chain 10 --steps 20 --warmup 10 --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain --no-other-configs > $R/gpurun_out/r05_bench_prof_1x640.json 2>/dev/null
cd $R
f=$(ls gpurun_out/prof_bench_1x640/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_bench_1x640_kernel_stats.csv; rm -rf gpurun_out/prof_bench_1x640
head -3 gpurun_out/r05_bench_1x640_kernel_stats.csv | cut -c1-150
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_prof_1x640.json')); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
