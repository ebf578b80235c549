cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
{ python tools/conv_once.py 32 384 128 128 5 1; for a in 0 1 2 3 4 6 7 16 23; do LGEN_WINO_ABLATE=$a python tools/conv_once.py 32 384 128 128 5 2; done; } 2>&1 | grep -v amdgpu > gpurun_out/r5_wino_ablate.log
cat gpurun_out/r5_wino_ablate.log
