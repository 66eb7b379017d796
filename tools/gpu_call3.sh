cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SVD_DEBUG_SYNC=1 AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
timeout 600 python -X faulthandler -m pytest "tests/test_gpu_ar_parity.py::test_autoregressive_chunks_vs_oracle" -m gpu -q -s -x > gpurun_out/r02_ar_debug2.log 2>&1
grep -v "^  File \"/usr" gpurun_out/r02_ar_debug2.log | head -30
