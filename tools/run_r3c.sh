R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_driver.py tests/test_gpu_gpt.py tests/test_gpu_vq.py -q -k "passes_bit_identical or driver or gptl or 24x24 or batches_per_chain or pipeline" > gpurun_out/r3c_tests.log 2>&1
tail -15 gpurun_out/r3c_tests.log
timeout 1200 python tools/exp_r3c.py > gpurun_out/exp_r3c.log 2>&1
tail -40 gpurun_out/exp_r3c.log
