# Full-depth parity of BASELINE configs 3-5 (GPT-XXL / GPT-3B / GPT-XL t2i) against the CPU oracle: minutes of host time per model.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out && rm -f gpurun_out/headline_parity.jsonl
LGEN_SLOW=1 timeout 2400 python -m pytest tests/test_gpu_headline.py -q -k full_depth > gpurun_out/slow_parity.log 2>&1
tail -5 gpurun_out/slow_parity.log
cp gpurun_out/headline_parity.jsonl gpurun_out/r03_full_depth_parity.jsonl 2>/dev/null
