run() { LGEN_ATTN_VARIANT=$1 timeout 900 python bench.py --no-cpu-baseline --no-live-traffic --no-solo --no-roofline --no-one-chain --batches-per-chain $2 --lanes $3 --steps $4 --warmup $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn variant $1 bpc $2 lanes $3 steps $4:', d['value'])"; }
run 13 8 2 32
run 12 8 2 32
run 10 16 2 32
run 10 12 2 24
run 10 8 2 16
run 10 4 2 32
run 10 4 3 24
