cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
(time timeout 1200 python tools/stress_kernels.py --iters 5000 --harness 200) > gpurun_out/r6_stress_final_normal.log 2>&1
tail -3 gpurun_out/r6_stress_final_normal.log
grep -c "'bad_launches': 0" gpurun_out/r6_stress_final_normal.log; grep -v "'bad_launches': 0" gpurun_out/r6_stress_final_normal.log | grep bad_launches | head
( time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r6_gpu_suite_final_second_box.log 2>&1
cat gpurun_out/r6_gpu_suite_final_second_box.log
