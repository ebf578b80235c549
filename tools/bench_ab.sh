#!/bin/bash
# Whole-bench A/B on ONE box (what decided every schedule / tile-shape choice of rounds 4-6: a launch's duration alone on the chip does not
# predict the two-chain pipeline -- DESIGN 4a items 6 and 10).  Each arm is "name:ENV=value[|ENV=value...]" (empty env list = the tree as it is);
# the arms run in the given order, so list them twice (a b a b) to see the box's own spread.
#   tools/bench_ab.sh "--config 4 --steps 8 --warmup 2" base: "old:LGEN_TILE_SHAPES=qkv=8,1,1,6,2,4,4;w13=4,1,2,6,2,4,4" base: ...
#   tools/bench_ab.sh "" v1:LGEN_CF_VARIANT=1 v3:LGEN_CF_VARIANT=3 v1:LGEN_CF_VARIANT=1 v3:LGEN_CF_VARIANT=3
# The round-6 logs under profiles/ (r06_c2_bench_ab, r06_c4_shape_ab[2], r06_c3_c5_shape_ab, r06_conv_pipe_ab, r06_dot2_ab, r06_widen[2]) were made this way.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
ARGS="$1"; shift
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline --allow-untested-schedule $ARGS"
for arm in "$@"; do
  name="${arm%%:*}"; envs="${arm#*:}"
  echo -n "$name: "
  ( IFS='|'; set -f; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; IFS=$' \t\n'; timeout 900 python bench.py $F 2>gpurun_out/bench_ab_err.log ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['batches_per_chain'], d['config']['chains_in_flight_per_gpu'])" || tail -3 gpurun_out/bench_ab_err.log
done 2>&1 | tee -a gpurun_out/bench_ab.log
