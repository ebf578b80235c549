cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_vq.py -q -x -k "conv_wino" 2>&1 | tail -12 ) > gpurun_out/r5_wino2_test.log 2>&1
tail -12 gpurun_out/r5_wino2_test.log
{ timeout 120 python tools/conv_once.py 32 384 128 128 5 1; for a in 0 1 2 4 16 23; do LGEN_WINO_ABLATE=$a timeout 120 python tools/conv_once.py 32 384 128 128 5 2; done; timeout 120 python tools/conv_once.py 32 192 256 256 5 1; timeout 120 python tools/conv_once.py 32 192 256 256 5 2; } 2>&1 | grep -v amdgpu > gpurun_out/r5_wino2_ablate.log
cat gpurun_out/r5_wino2_ablate.log
( LGEN_VQ_WINO=1 timeout 600 python -m pytest tests/test_gpu_vq.py -q -x 2>&1 | tail -5 ) > gpurun_out/r5_wino2_vq_tests.log 2>&1
tail -3 gpurun_out/r5_wino2_vq_tests.log
( LGEN_VQ_WINO=1 timeout 300 python tools/vq_once.py 32 5 2>&1 | tail -1 )
