#!/bin/bash
# round 6: reproduce / classify the intermittent wrong result of GPUTEST_r05 (see tools/stress_kernels.py)
mkdir -p gpurun_out
ITERS=${ITERS:-3000}
(time python tools/stress_kernels.py --iters $ITERS --harness 200) > gpurun_out/r6_stress_normal.log 2>&1
tail -3 gpurun_out/r6_stress_normal.log
(time python tools/stress_kernels.py --iters $ITERS --fresh --hammer --harness 200) > gpurun_out/r6_stress_fresh_hammer.log 2>&1
tail -3 gpurun_out/r6_stress_fresh_hammer.log
for i in $(seq 1 ${LOOPS:-12}); do
  python -m pytest tests/test_gpu_gpt.py -q -x -k "test_qkv_rope_append_and_attention and 192-16" 2>&1 | tail -3 >> gpurun_out/r6_pytest_loop.log
done
grep -c passed gpurun_out/r6_pytest_loop.log; grep -i -B2 -A2 "failed\|error" gpurun_out/r6_pytest_loop.log | tail -20
