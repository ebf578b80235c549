# round 5: the bench with the VQ decoder's 3x3 convolutions in Winograd form (opt-in) against the direct form, alternating on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline"
for i in 1 2; do for w in 0 1; do
  echo -n "LGEN_VQ_WINO=$w run $i: "; LGEN_VQ_WINO=$w timeout 200 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r5_wino_bench_ab.log
