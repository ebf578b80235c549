# attention key-load clamp, same-box A/B in the bench (alternating), VQ on the direct form
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs --no-roofline"
export LGEN_VQ_WINO=0
for i in 1 2 3; do
  for c in 0 1; do
    LGEN_ATTN_CLAMP=$c timeout 200 python bench.py $F > gpurun_out/r5_ab_clamp${c}_$i.json 2>/dev/null
    echo -n "clamp=$c run $i: "; python -c "import json; d=json.load(open('gpurun_out/r5_ab_clamp${c}_$i.json')); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r5_attn_clamp_ab.log
