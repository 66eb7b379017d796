cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r02_gemm_tests3.log 2>&1; tail -3 gpurun_out/r02_gemm_tests3.log
timeout 300 python tools/gemm_probe2.py 8,2,1,18,19,20 5 > gpurun_out/r02_gemm_probe_newloop.log 2>&1; cut -c1-200 gpurun_out/r02_gemm_probe_newloop.log | grep -v "^/opt"
timeout 900 python tools/tune_gemm.py > gpurun_out/r02_tune.log 2>&1; tail -50 gpurun_out/r02_tune.log
timeout 900 python bench.py --steps 7 --warmup 1 > gpurun_out/r02_bench_newloop.json 2> gpurun_out/r02_bench_newloop.err; head -c 700 gpurun_out/r02_bench_newloop.json; echo; tail -3 gpurun_out/r02_bench_newloop.err
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_suite.log 2>&1; tail -8 gpurun_out/r02_gpu_suite.log
