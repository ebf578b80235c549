# First GPU call of round 5: are the prepared tile shapes (r5-prep) right, do they pay, and does the L2 touch-ahead of the loader
# waves (LGEN_TILE_TOUCH=1) shorten the GEMMs beside another chain's attention stream?
#   gpurun --timeout 1500 -- 'bash tools/run_r5_first.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
# 1. correctness: every shape (incl. lw 12 / 16, the deeper rings, the touch-ahead) must agree BIT for bit with the others, the first with the oracle
( LGEN_TILE_TOUCH=1 timeout 400 python -m pytest tests/test_gpu_headline.py -q -x -k "tile_gemm_family" 2>&1 | tail -15 ) > gpurun_out/r5_tile_tests.log 2>&1
# 2. alone on the chip: us per launch of every shape at 640 rows (GPT-L), without / with the touch-ahead
( LGEN_TILE_TOUCH=0 timeout 300 python tools/gemm_tile_sweep.py GPT-L 640 2>&1 | grep -v "^$" ) > gpurun_out/r5_sweep_L_t0.log 2>&1
( LGEN_TILE_TOUCH=1 timeout 300 python tools/gemm_tile_sweep.py GPT-L 640 2>&1 | grep -v "^$" ) > gpurun_out/r5_sweep_L_t1.log 2>&1
# 3. beside the other chain's attention: how far the GEMM graph stretches
( NO_V=1 ROWS=640 LGEN_TILE_TOUCH=0 timeout 300 python tools/overlap_probe.py 2>&1 | grep -v "^$" ) > gpurun_out/r5_overlap_t0.log 2>&1
( NO_V=1 ROWS=640 LGEN_TILE_TOUCH=1 timeout 300 python tools/overlap_probe.py 2>&1 | grep -v "^$" ) > gpurun_out/r5_overlap_t1.log 2>&1
# 4. in the bench (the only judge)
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-roofline --allow-untested-schedule"
LN="w13=2,2,4,4,2,4,16;head=2,2,4,4,2,4,16"
DEEP="qkv=4,1,1,8,2,6,4;wo=2,2,2,2,2,9,4;w2=2,2,2,2,2,9,4"
( LGEN_TILE_TOUCH=0 timeout 200 python bench.py $F ) > gpurun_out/r5_ab_default1.json 2>gpurun_out/r5_ab_default1.err
( LGEN_TILE_TOUCH=1 timeout 200 python bench.py $F ) > gpurun_out/r5_ab_touch1.json 2>gpurun_out/r5_ab_touch1.err
( LGEN_TILE_TOUCH=0 LGEN_TILE_SHAPES="$LN" timeout 200 python bench.py $F ) > gpurun_out/r5_ab_ln1.json 2>gpurun_out/r5_ab_ln1.err
( LGEN_TILE_TOUCH=0 LGEN_TILE_SHAPES="$DEEP" timeout 200 python bench.py $F ) > gpurun_out/r5_ab_deep1.json 2>gpurun_out/r5_ab_deep1.err
( LGEN_TILE_TOUCH=1 timeout 200 python bench.py $F ) > gpurun_out/r5_ab_touch2.json 2>gpurun_out/r5_ab_touch2.err
( LGEN_TILE_TOUCH=0 timeout 200 python bench.py $F ) > gpurun_out/r5_ab_default2.json 2>gpurun_out/r5_ab_default2.err
tail -4 gpurun_out/r5_tile_tests.log
for f in default1 touch1 ln1 deep1 touch2 default2; do echo -n "$f "; head -c 110 gpurun_out/r5_ab_$f.json | tail -c 50; echo; done
cat gpurun_out/r5_overlap_t0.log gpurun_out/r5_overlap_t1.log
grep -E "MISMATCH|rc " gpurun_out/r5_sweep_L_t0.log gpurun_out/r5_sweep_L_t1.log | head
