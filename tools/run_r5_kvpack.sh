# round 5: K/V rows packed at head_dim rounded up to 16 bytes (GPT-3B: 104 instead of 128 elements) -- parity, then config 4 A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_gpt.py -q -x -k "qkv_rope_append_and_attention or hd100 or prefill or whole_sequence" 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r5_kvpack_tests1.log
( timeout 900 python -m pytest tests/test_gpu_headline.py -q -x -k "config4 or gpt3b or tile_gemm_family" 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r5_kvpack_tests2.log
( timeout 600 python -m pytest tests/test_gpu_serve.py tests/test_gpu_gpt.py -q -x -k "serve or golden" 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r5_kvpack_tests3.log
F="--config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic --no-solo --no-one-chain"
for i in 1 2; do for c in 0 1; do
  echo -n "config 4, LGEN_KV_PACK=$c run $i: "; LGEN_KV_PACK=$c timeout 300 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done 2>&1 | tee gpurun_out/r5_kvpack_ab.log
