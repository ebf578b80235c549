cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
{ LGEN_WINO_ABLATE=32 timeout 120 python tools/conv_once.py 32 384 128 128 3 2; LGEN_WINO_ABLATE=0 timeout 120 python tools/conv_once.py 32 384 128 128 3 2; } 2>&1 | grep -v amdgpu | tee gpurun_out/r5_wino_time.log
