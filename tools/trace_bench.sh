# rocprofv3 kernel trace of the headline bench -> overlap summary (tools/trace_overlap.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/trace_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -- python $R/bench.py --no-cpu-baseline --no-live-traffic --no-solo --no-roofline --steps 12 --warmup 4 > $R/gpurun_out/trace_bench.log 2>&1
F=$(find /tmp/trace_out -name "*kernel_trace.csv" | head -1)
ls -la $F >> $R/gpurun_out/trace_bench.log
python $R/tools/trace_overlap.py $F 0.35 0.9 > $R/gpurun_out/trace_overlap.json 2>> $R/gpurun_out/trace_bench.log
