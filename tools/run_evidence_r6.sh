# Round-6 evidence run (gpurun): PMC passes (FETCH_SIZE / WRITE_SIZE, separate) at the bench's chain shape (640 rows), the bench line as the driver
# runs it (with other_configs), rocprofv3 kernel stats of the bench itself (default two-chain schedule; ONE 640-row chain in flight).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.log 2>&1
cp profiles/r06_pmc.json profiles/r06_pmc.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
( time timeout 1500 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err ) 2> gpurun_out/r06_bench_line.time
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_default -- python $R/bench.py --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain --no-other-configs > $R/gpurun_out/r06_bench_prof_default.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_1x640 -- python $R/bench.py --lanes 1 --steps 10 --warmup 10 --batches-per-chain 10 --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain --no-other-configs > $R/gpurun_out/r06_bench_prof_1x640.json 2>/dev/null
cd $R
for d in prof_bench_default prof_bench_1x640; do f=$(ls gpurun_out/$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r06_${d#prof_}_kernel_stats.csv; rm -rf gpurun_out/$d; done
head -c 400 gpurun_out/r06_bench_line.json; echo; tail -2 gpurun_out/r06_bench_line.err; cat gpurun_out/r06_bench_line.time; tail -14 gpurun_out/pmc_summary.log
