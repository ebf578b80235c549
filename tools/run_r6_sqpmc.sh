#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/tools/pmc_sq_target.py > $R/gpurun_out/pmc_sq.log 2>&1
tail -3 $R/gpurun_out/pmc_sq.log
cd $R && python tools/pmc_sq_summary.py gpurun_out/pmc_sq --json gpurun_out/r06_sq_pmc.json > gpurun_out/r06_sq_pmc.csv 2> gpurun_out/pmc_sq_summary.err
tail -3 gpurun_out/pmc_sq_summary.err; head -c 3000 gpurun_out/r06_sq_pmc.csv
ls gpurun_out/pmc_sq/*/ | head; rm -rf gpurun_out/pmc_sq
