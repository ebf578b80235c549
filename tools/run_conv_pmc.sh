cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in 0 1; do python $R/tools/conv_once.py 16 384 128 128 5 $v 2>&1 | grep conv_fused; done
python $R/tools/conv_once.py 16 192 256 256 5 0 2>&1 | grep conv_fused
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/conv_pmc1 -- python $R/tools/conv_once.py 16 384 128 128 3 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/conv_pmc2 -- python $R/tools/conv_once.py 16 384 128 128 3 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/conv_pmc3 -- python $R/tools/conv_once.py 16 384 128 128 3 0 > /dev/null 2>&1
cd $R && python - <<'PY'
import csv, glob, collections
for d in ("conv_pmc1","conv_pmc2","conv_pmc3"):
    agg=collections.defaultdict(float); n=0
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(d, {k: f"{v:.4g}" for k,v in agg.items()})
PY
