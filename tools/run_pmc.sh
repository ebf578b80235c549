cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.log 2>&1
cp profiles/r02_pmc.json profiles/r02_pmc.csv gpurun_out/ 2>/dev/null
python bench.py --steps 6 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -c 3000 gpurun_out/bench_a.json; tail -5 gpurun_out/bench_a.err; tail -20 gpurun_out/pmc_summary.log
