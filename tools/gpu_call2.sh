cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1
timeout 600 python -m pytest tests/test_gpu_ar_parity.py -m gpu -q -s -x > gpurun_out/r02_ar_debug.log 2>&1
grep -v "^  File \"/usr" gpurun_out/r02_ar_debug.log | head -40
unset AMD_SERIALIZE_KERNEL HIP_LAUNCH_BLOCKING PYTORCH_NO_CUDA_MEMORY_CACHING
timeout 600 python -m pytest tests/test_gpu_kernels_product_sizes.py -m gpu -q -s > gpurun_out/r02_product_tests.log 2>&1
grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/r02_product_tests.log | tail -40
