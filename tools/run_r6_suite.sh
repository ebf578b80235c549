#!/bin/bash
# the whole GPU suite + smoke() on the tree as it stands (round 6)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r6_gpu_suite_final.log 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) >> gpurun_out/r6_gpu_suite_final.log
cat gpurun_out/r6_gpu_suite_final.log
