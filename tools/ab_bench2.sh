# chain width x chains in flight with the tile family (and the skinny kernels for reference)
run() { LGEN_GEMM_TILE=$1 timeout 900 python bench.py --no-cpu-baseline --no-live-traffic --no-solo --no-roofline --no-one-chain --batches-per-chain $2 --lanes $3 --steps $4 --warmup $5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile=$1 bpc=$2 lanes=$3 steps=$4', d['value'])" || echo "tile=$1 bpc=$2 lanes=$3 failed"; }
run 1 4 3 24 4
run 1 8 2 32 8
run 1 8 3 24 8
run 1 16 1 32 16
run 1 16 2 32 16
run 0 8 2 32 8
run 0 16 2 32 16
