# Round-3 diagnostic run #1 (gpurun): GEMM tile sweep at 64/128/256 rows, SQ counters of the decode-chain kernels at 128 rows,
# wide-chain pipeline points.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python tools/gemm_sweep.py 64 128 256 > gpurun_out/gemm_sweep.log 2>&1
tail -5 gpurun_out/gemm_sweep.log
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_sq1 -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_sq1.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_sq2.log 2>&1
cd $R
python tools/pmc_sq_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 > gpurun_out/pmc_sq_summary.csv 2> gpurun_out/pmc_sq_summary.err
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
head -30 gpurun_out/pmc_sq_summary.csv
timeout 900 python tools/exp_r3a.py > gpurun_out/exp_r3a.log 2>&1
tail -30 gpurun_out/exp_r3a.log
