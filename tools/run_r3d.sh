R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 300 python tools/diag_m256.py > gpurun_out/diag_m256.log 2>&1; cat gpurun_out/diag_m256.log | tail -30
timeout 600 python tools/gemm_sweep.py 128 256 > gpurun_out/gemm_sweep3.log 2>&1
grep "db 1\|best" gpurun_out/gemm_sweep3.log | grep "2, 4, 8\|best" | head -30
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r3d.json 2> gpurun_out/bench_r3d.err
tail -c 2500 gpurun_out/bench_r3d.json; tail -5 gpurun_out/bench_r3d.err
