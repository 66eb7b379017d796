cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r02_gemm_tests.log 2>&1; tail -3 gpurun_out/r02_gemm_tests.log
timeout 600 python -m pytest tests/test_gpu_ar_parity.py -m gpu -q -s > gpurun_out/r02_ar_tests.log 2>&1; grep -E "^\[|passed|failed|rror" gpurun_out/r02_ar_tests.log | tail
timeout 600 python tools/gemm_probe2.py 8,21,22,25,2,23,1,24 7 > gpurun_out/r02_gemm_probe2.log 2>&1; cat gpurun_out/r02_gemm_probe2.log
