# round 5: persistent-attention variants after the key-load clamp, and the deeper GEMM rings, same-box A/B in the bench
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule"
DEEP="qkv=4,1,1,8,2,6,4;wo=2,2,2,2,2,9,4;w2=2,2,2,2,2,9,4"
DEEP2="wo=2,2,2,2,2,6,4;w2=2,2,2,2,2,6,4"
run() { echo -n "$1: "; env $2 timeout 200 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
{
run default "X=1"
run attn8 "LGEN_ATTN_VARIANT=8"
run attn14 "LGEN_ATTN_VARIANT=14"
run deep "LGEN_TILE_SHAPES=$DEEP"
run default "X=1"
run attn11 "LGEN_ATTN_VARIANT=11"
run deep2 "LGEN_TILE_SHAPES=$DEEP2"
run deep "LGEN_TILE_SHAPES=$DEEP"
run default "X=1"
} 2>&1 | tee gpurun_out/r5_ab3.log
