# Round 5, second GPU call: the new parity tests, the default bench line (with other_configs), a kernel profile of decode_code().
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/headline_parity.jsonl
( timeout 1500 python -m pytest tests/test_gpu_headline.py -q -x --durations=12 -k "graph_replay_equals or same_batch_alone or free_running or ten_batches or config3_gptxxl or config4_gpt3b or full_depth or tile_gemm_family" 2>&1 | tail -30 ) > gpurun_out/r5_tests2.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r5_bench_try1.json 2> gpurun_out/r5_bench_try1.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vq -- python $R/tools/vq_once.py 32 3 > $R/gpurun_out/r5_vq_once.log 2>&1
cd $R
f=$(ls gpurun_out/prof_vq/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_vq_decode_kernel_stats.csv
f=$(ls gpurun_out/prof_vq/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python - "$f" <<'P' > gpurun_out/r05_vq_decode_trace_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last decode_code() call: kernels after the last lookup_pqconv
idx = [i for i, r in enumerate(rows) if "lookup_pqconv" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  grid {r.get("Grid_Size_X", "?"):>8} wg {r.get("Workgroup_Size_X", "?"):>4}  {r["Kernel_Name"][:90]}')
P
rm -rf gpurun_out/prof_vq
tail -25 gpurun_out/r5_tests2.log; head -c 400 gpurun_out/r5_bench_try1.json; echo; tail -3 gpurun_out/r5_bench_try1.err
python -c "
import json; d=json.load(open('gpurun_out/r5_bench_try1.json')); print({k: d[k] for k in ('value','images_per_s_with_one_chain_in_flight','images_per_s_with_one_step_in_flight') if k in d}); print(d.get('other_configs')); print(d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_gemm']['us_per_step'], d['roofline_vq_decode']['ms_per_decode_code'])"
