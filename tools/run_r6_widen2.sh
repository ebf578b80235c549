#!/bin/bash
# round 6: the wide-chain schedules of configs 3 / 4 / 5 with their measured tile tables: oracle tests, then the bench lines; then the SQ counter pass again (fixed summary)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_headline.py -q -x -k "wide_chain or config3 or config4 or config5" 2>&1 | tail -5 ) | tee gpurun_out/r6_widen2_tests.log
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline"
run() { echo -n "$1: "; shift; timeout 600 python bench.py $F "$@" 2>gpurun_out/r6_widen2_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['batches_per_chain'], d['config']['chains_in_flight_per_gpu'], d['gemm_schedule']['tested_end_to_end'], d['gemm_schedule']['schedule'])" || tail -3 gpurun_out/r6_widen2_err.log; }
{
run c3 --config 3 --steps 16 --warmup 4
run c4 --config 4 --steps 8 --warmup 2
run c5 --config 5 --steps 16 --warmup 4
run c5_2x12 --config 5 --steps 24 --warmup 4 --batches-per-chain 12 --lanes 2
run c3 --config 3 --steps 16 --warmup 4
run c4 --config 4 --steps 8 --warmup 2
run c5 --config 5 --steps 16 --warmup 4
} 2>&1 | tee gpurun_out/r6_widen2.log
bash tools/run_r6_sqpmc.sh > gpurun_out/r6_sqpmc_run.log 2>&1
head -c 1500 gpurun_out/r06_sq_pmc.csv
