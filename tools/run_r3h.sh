R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "passes_bit_identical" > gpurun_out/r3h_tests.log 2>&1; tail -3 gpurun_out/r3h_tests.log
timeout 600 python tools/gemm_sweep.py 512 > gpurun_out/gemm_sweep512.log 2>&1; grep "best" gpurun_out/gemm_sweep512.log
timeout 900 python tools/exp_r3d.py > gpurun_out/exp_r3d.log 2>&1; grep R3D gpurun_out/exp_r3d.log; tail -3 gpurun_out/exp_r3d.log | grep -v R3D
