timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -k "qkv_rope_append_and_attention" 2>&1 | grep -E "Error|assert|FAILED" | head -8
