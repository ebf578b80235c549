cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_gpt.py -q -k "qkv_rope_append_and_attention or golden or prefill or whole_sequence" 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r5_kvpack_tests1.log
( timeout 600 python -m pytest tests/test_gpu_serve.py tests/test_gpu_driver.py -q 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r5_kvpack_tests3.log
