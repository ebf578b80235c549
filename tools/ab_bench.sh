# A/B of the headline bench: skinny kernels (LGEN_GEMM_TILE=0) against the big-M tile family (default)
for t in 0 1 0 1; do
  LGEN_GEMM_TILE=$t timeout 600 python bench.py --no-cpu-baseline --no-live-traffic --no-solo --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile=$t', d['value'], d.get('images_per_s_with_one_chain_in_flight'), d.get('roofline_gemm',{}).get('us_per_step'), d.get('roofline',{}).get('frac'))"
done
