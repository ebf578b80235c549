#!/bin/bash
# round 6: after the RoPE fix -- the failing test in a loop, the long stress in three modes, then the whole GPU suite
mkdir -p gpurun_out
rm -f gpurun_out/r6_pytest_loop_fixed.log
for i in $(seq 1 ${LOOPS:-12}); do
  python -m pytest tests/test_gpu_gpt.py -q -x -k "test_qkv_rope_append_and_attention and 192-16" 2>&1 | tail -2 >> gpurun_out/r6_pytest_loop_fixed.log
done
grep -c passed gpurun_out/r6_pytest_loop_fixed.log; grep -ci "failed" gpurun_out/r6_pytest_loop_fixed.log
(time python tools/stress_kernels.py --iters ${ITERS:-5000} --harness 200) > gpurun_out/r6_stress_fixed_normal.log 2>&1; tail -2 gpurun_out/r6_stress_fixed_normal.log | head -1
(time python tools/stress_kernels.py --iters ${ITERS:-5000} --fresh --hammer) > gpurun_out/r6_stress_fixed_fresh_hammer.log 2>&1; grep STRESS gpurun_out/r6_stress_fixed_fresh_hammer.log
(time AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 python tools/stress_kernels.py --iters 1000) > gpurun_out/r6_stress_fixed_serialized.log 2>&1; grep STRESS gpurun_out/r6_stress_fixed_serialized.log
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r6_gpu_suite.log 2>&1; tail -5 gpurun_out/r6_gpu_suite.log
