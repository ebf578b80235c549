R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for rows in 64 128 256; do ROWS=$rows timeout 300 python tools/attn_sweep.py 2 6 7 2>&1 | grep variant | sed "s/^/rows $rows /" | cut -c1-150; done | tee gpurun_out/attn_sweep_r3.log
timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_serve.py tests/test_gpu_gpt.py tests/test_gpu_driver.py -q -k "passes_bit_identical or serve or requests or staggered or qkv_rope_append_and_attention or driver" > gpurun_out/r3g_tests.log 2>&1
tail -8 gpurun_out/r3g_tests.log
timeout 600 python tools/serve_bench.py 32 64 > gpurun_out/serve_bench.log 2>&1; grep -v amdgpu gpurun_out/serve_bench.log | tail -6
