timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -k "qkv_rope_append_and_attention" 2>&1 | tail -2
for v in 6 8 10 11 12 13; do echo "== attn variant $v"; LGEN_ATTN_VARIANT=$v ROWS=256 timeout 300 python tools/overlap_probe.py 2>&1 | grep -E "alone|A\|G|A\|A"; done
