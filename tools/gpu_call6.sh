cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r02_gemm_tests2.log 2>&1; tail -3 gpurun_out/r02_gemm_tests2.log
timeout 600 python tools/gemm_probe2.py 8,2,1,18 5 > gpurun_out/r02_gemm_probe_lean.log 2>&1; cat gpurun_out/r02_gemm_probe_lean.log
timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_lean.json 2> gpurun_out/r02_bench_lean.err; head -c 600 gpurun_out/r02_bench_lean.json; tail -3 gpurun_out/r02_bench_lean.err
