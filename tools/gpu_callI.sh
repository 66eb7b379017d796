cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r02_kernels_newloop3.log 2>&1
tail -3 gpurun_out/r02_kernels_newloop3.log
timeout 900 python tools/tune_gemm.py > gpurun_out/r02_tune4.log 2>&1
grep -E "GEMM time|wrote" gpurun_out/r02_tune4.log
cp streamingt2v_amd/gemm_tiles.json gpurun_out/gemm_tiles_r02d.json
timeout 600 python bench.py --steps 6 --warmup 1 > gpurun_out/r02_bench_newloop2.json 2> gpurun_out/r02_bench_newloop2.err
cut -c1-300 gpurun_out/r02_bench_newloop2.json
