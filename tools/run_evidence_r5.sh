# Round-5 evidence run (gpurun): the bench line (with other_configs), rocprofv3 kernel stats of the bench itself (default schedule,
# one chain in flight), config 3, PMC passes at the bench's chain shape (640 rows).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.log 2>&1
cp profiles/r05_pmc.json profiles/r05_pmc.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_default -- python $R/bench.py --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain --no-other-configs > $R/gpurun_out/r05_bench_prof_default.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_l1 -- python $R/bench.py --lanes 1 --steps 10 --warmup 10 --no-cpu-baseline --no-solo --no-live-traffic --no-one-chain --no-other-configs > $R/gpurun_out/r05_bench_prof_lanes1.json 2>/dev/null
cd $R
for d in prof_bench_default prof_bench_l1; do f=$(ls gpurun_out/$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_${d}_kernel_stats.csv; rm -rf gpurun_out/$d; done
timeout 300 python bench.py --config 3 --no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --steps 12 --warmup 4 > gpurun_out/r05_bench_config3.json 2> gpurun_out/r05_bench_config3.err
head -c 300 gpurun_out/r05_bench_line.json; echo; tail -2 gpurun_out/r05_bench_line.err; head -c 200 gpurun_out/r05_bench_config3.json; echo; tail -14 gpurun_out/pmc_summary.log
