# Round-end validation on the GPU box: full GPU suite, smoke, the driver's bench line, its rocprof kernel-trace summary, PMC traffic of the top GEMM signatures.
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r02_final_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r02_final_gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_stage1_final.json 2> gpurun_out/r02_bench_stage1_final.err; cut -c1-220 gpurun_out/r02_bench_stage1_final.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_final_kt.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_kt -name "*.db" | head -1) $R/gpurun_out/r02_stage1_kernel_stats.txt > /dev/null; rm -rf $R/gpurun_out/prof_kt
head -12 $R/gpurun_out/r02_stage1_kernel_stats.txt | cut -c1-180
cd $R
rm -f gpurun_out/r02_traffic_signatures.json
for S in m0_M460800_N2560_K320_s0_u0_e1_o0 m0_M460800_N320_K1280_s0_u0_e0_o0 m0_M115200_N5120_K640_s0_u0_e1_o0; do bash tools/pmc_signature.sh $S $R/gpurun_out/r02_traffic_signatures.json; done
timeout 300 python bench.py --workload c2 --steps 3 --warmup 1 > gpurun_out/r02_bench_c2_fp16.json 2>/dev/null; cut -c1-160 gpurun_out/r02_bench_c2_fp16.json
timeout 300 python bench.py --workload enhance --steps 1 --warmup 1 > gpurun_out/r02_bench_enhance.json 2>/dev/null; cut -c1-160 gpurun_out/r02_bench_enhance.json
