# round 5: the whole GPU suite on the current tree (the log is committed as profiles/r05_gpu_tests.log)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; rm -f gpurun_out/headline_parity.jsonl
( timeout 2400 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40 ) > gpurun_out/r05_gpu_tests.log 2>&1
tail -25 gpurun_out/r05_gpu_tests.log
