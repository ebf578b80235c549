# Round-3 run #2 (gpurun): parity of the persistent fused-norm GEMMs, then tile x passes sweep, then pipeline points
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -k "passes_bit_identical or four_batches or two_batches or b64_logits or qkv_fused" > gpurun_out/r3b_tests.log 2>&1
tail -5 gpurun_out/r3b_tests.log
timeout 900 python tools/gemm_sweep.py 64 128 256 > gpurun_out/gemm_sweep2.log 2>&1
grep "best LGEN" gpurun_out/gemm_sweep2.log
timeout 900 python tools/exp_r3a.py > gpurun_out/exp_r3b.log 2>&1
tail -16 gpurun_out/exp_r3b.log
