# First GPU call of round 5 (branch r5-prep): are the prepared tile shapes right, and do they pay?  ~8 GPU-minutes.
#   gpurun --timeout 900 -- 'bash tools/run_r5_first.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
# 1. correctness: every shape (incl. lw 12 / 16 and the deeper rings) must agree BIT for bit with the others, the first with the oracle
( timeout 300 python -m pytest tests/test_gpu_headline.py -q -x -k "tile_gemm_family" 2>&1 | tail -15 ) > gpurun_out/r5_tile_tests.log 2>&1
# 2. alone on the chip: us per launch of every shape, 640 and 256 rows (GPT-L), 256 rows (GPT-3B: the batched statistics loads)
( timeout 300 python tools/gemm_tile_sweep.py GPT-L 640 256 2>&1 | grep -v "^$" ) > gpurun_out/r5_sweep_L.log 2>&1
( LGEN_SWEEP_KINDS=qkv,w13,head timeout 300 python tools/gemm_tile_sweep.py GPT-3B 256 2>&1 | grep -v "^$" ) > gpurun_out/r5_sweep_3B.log 2>&1
# 3. in the bench (the only judge): default schedule against the candidates, alternating
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-roofline --allow-untested-schedule"
LN="w13=2,2,4,4,2,4,16;head=2,2,4,4,2,4,16"
DEEP="qkv=4,1,1,8,2,6,4;wo=2,2,2,2,2,9,4;w2=2,2,2,2,2,9,4"
for i in 1 2; do
  ( timeout 200 python bench.py $F ) > gpurun_out/r5_ab_default$i.json 2>/dev/null
  ( LGEN_TILE_SHAPES="$LN" timeout 200 python bench.py $F ) > gpurun_out/r5_ab_ln$i.json 2>/dev/null
  ( LGEN_TILE_SHAPES="$DEEP" timeout 200 python bench.py $F ) > gpurun_out/r5_ab_deep$i.json 2>/dev/null
done
tail -4 gpurun_out/r5_tile_tests.log
for f in default1 ln1 deep1 default2 ln2 deep2; do echo -n "$f "; head -c 110 gpurun_out/r5_ab_$f.json | tail -c 40; echo; done
grep -E "w13|head" gpurun_out/r5_sweep_L.log | grep -E "rows 640" | grep -E ", 16\)|8, 1, 1, 8|skinny" | cut -c1-100
