# Round-2 evidence run (gpurun): configs 3-5 at full size, bench line, rocprofv3 kernel stats of the bench (default lanes and 1 lane)
R=/root/repo
cd $R
python tools/smoke_configs.py > gpurun_out/r02_configs_3_4_5.log 2>&1
python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_default -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r02_bench_prof_default.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_l1 -- python $R/bench.py --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02_bench_prof_lanes1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vq -- python $R/tools/vq_once.py 32 3 > $R/gpurun_out/r02_vq_once.log 2>/dev/null
cd $R
for d in prof_bench_default prof_bench_l1 prof_vq; do f=$(ls gpurun_out/$d/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r02_${d}_kernel_stats.csv; rm -f gpurun_out/$d/*/*kernel_trace.csv; done
grep -v amdgpu gpurun_out/r02_configs_3_4_5.log; head -c 600 gpurun_out/r02_bench_line.json; echo; head -5 gpurun_out/r02_prof_bench_l1_kernel_stats.csv | cut -c1-150
