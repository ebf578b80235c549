# Round-4 evidence run (gpurun): PMC passes at the bench's chain shape (640 rows), the bench line, rocprofv3 kernel stats of the
# bench itself (default schedule) and of one chain in flight.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.log 2>&1
cp profiles/r04_pmc.json profiles/r04_pmc.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_default -- python $R/bench.py --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain > $R/gpurun_out/r04_bench_prof_default.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_l1 -- python $R/bench.py --lanes 1 --steps 10 --warmup 10 --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain > $R/gpurun_out/r04_bench_prof_lanes1.json 2>/dev/null
cd $R
for d in prof_bench_default prof_bench_l1; do f=$(ls gpurun_out/$d/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r04_${d}_kernel_stats.csv; rm -rf gpurun_out/$d; done
head -c 400 gpurun_out/r04_bench_line.json; echo; tail -3 gpurun_out/r04_bench_line.err; head -8 gpurun_out/r04_prof_bench_l1_kernel_stats.csv | cut -c1-150; tail -40 gpurun_out/pmc_summary.log
