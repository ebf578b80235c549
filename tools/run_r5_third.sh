# Round 5, third GPU call: the Winograd conv kernel -- parity, end-to-end VQ tests, timing against the direct form.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_vq.py -q -x -k "conv_wino" 2>&1 | tail -15 ) > gpurun_out/r5_wino_test.log 2>&1
tail -5 gpurun_out/r5_wino_test.log
( timeout 900 python -m pytest tests/test_gpu_vq.py -q 2>&1 | tail -15 ) > gpurun_out/r5_vq_tests.log 2>&1
tail -5 gpurun_out/r5_vq_tests.log
( LGEN_VQ_WINO=0 timeout 300 python tools/vq_once.py 32 5 2>&1 | tail -3 ) > gpurun_out/r5_vq_once_direct.log 2>&1
( LGEN_VQ_WINO=1 timeout 300 python tools/vq_once.py 32 5 2>&1 | tail -3 ) > gpurun_out/r5_vq_once_wino.log 2>&1
cat gpurun_out/r5_vq_once_direct.log gpurun_out/r5_vq_once_wino.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vqw -- python $R/tools/vq_once.py 32 3 > /dev/null 2>&1
cd $R
f=$(ls gpurun_out/prof_vqw/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_vq_decode_wino_kernel_stats.csv && head -8 $f | cut -c1-150
rm -rf gpurun_out/prof_vqw
# attention: key loads clamped to the last visible key (gpt_ops.hip) -- parity of every variant, then the bench
( timeout 600 python -m pytest tests/test_gpu_gpt.py -q -x -k "attention or attn" 2>&1 | tail -6 ) > gpurun_out/r5_attn_tests.log 2>&1
tail -3 gpurun_out/r5_attn_tests.log
F="--no-cpu-baseline --no-live-traffic --no-solo --no-one-chain --no-other-configs"
( LGEN_VQ_WINO=0 timeout 300 python bench.py $F ) > gpurun_out/r5_bench_clamp_direct.json 2> gpurun_out/r5_bench_clamp_direct.err
( LGEN_VQ_WINO=1 timeout 300 python bench.py $F ) > gpurun_out/r5_bench_clamp_wino.json 2> gpurun_out/r5_bench_clamp_wino.err
for f in clamp_direct clamp_wino; do python -c "
import json; d=json.load(open('gpurun_out/r5_bench_$f.json')); print('$f', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_gemm']['us_per_step'], d['roofline_vq_decode']['ms_per_decode_code'], d['roofline']['launch_us_by_position'])"; done
