# Round-3 evidence run for the final tree (gpurun): PMC passes at the bench's chain shape (256 rows), the bench line, rocprofv3
# kernel stats of the bench itself (default schedule and one chain in flight), SQ counters of the decode-chain kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.log 2>&1
cp profiles/r03_pmc.json profiles/r03_pmc.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_default -- python $R/bench.py --no-cpu-baseline --no-solo --no-live-traffic > $R/gpurun_out/r03_bench_prof_default.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_l1 -- python $R/bench.py --lanes 1 --steps 4 --warmup 4 --no-cpu-baseline --no-solo --no-live-traffic > $R/gpurun_out/r03_bench_prof_lanes1.json 2>/dev/null
cd $R
for d in prof_bench_default prof_bench_l1; do f=$(ls gpurun_out/$d/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r03_${d}_kernel_stats.csv; rm -rf gpurun_out/$d; done
head -c 600 gpurun_out/r03_bench_line.json; echo; tail -3 gpurun_out/r03_bench_line.err; head -8 gpurun_out/r03_prof_bench_l1_kernel_stats.csv | cut -c1-150; tail -30 gpurun_out/pmc_summary.log
