cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02_gemm_tests4.log 2>&1; tail -3 gpurun_out/r02_gemm_tests4.log
timeout 300 python tools/gemm_probe2.py 8,20,18,1 5 > gpurun_out/r02_gemm_probe_epi.log 2>&1; cut -c1-215 gpurun_out/r02_gemm_probe_epi.log | grep -v "^/opt"
timeout 900 python bench.py --steps 7 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_epi.json 2> gpurun_out/r02_bench_epi.err; head -c 300 gpurun_out/r02_bench_epi.json; echo; tail -3 gpurun_out/r02_bench_epi.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --denoise-steps 2 --no-cpu-baseline --no-trace > $R/gpurun_out/r02_kt2.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_kt -name "*.db" | head -1) $R/gpurun_out/r02_stage1_2steps_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/prof_kt
head -24 $R/gpurun_out/r02_stage1_2steps_kernel_stats.txt | cut -c1-190
bash $R/tools/pmc_signature.sh m0_M460800_N2560_K320_s0_u0_e1_o0 $R/gpurun_out/r02_traffic_signatures.json
bash $R/tools/pmc_signature.sh m0_M460800_N320_K1280_s0_u0_e0_o0 $R/gpurun_out/r02_traffic_signatures.json
