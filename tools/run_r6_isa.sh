#!/bin/bash
mkdir -p gpurun_out
E=tools/experiments/r6_waw
python tools/isa_run_qkv.py --iters ${ITERS:-2000} $E/orig.hsaco $E/[J-N]_*.hsaco > gpurun_out/r6_isa_ab3.log 2>&1
cut -c1-330 gpurun_out/r6_isa_ab3.log
