#!/bin/bash
# Round-end validation on the GPU box (round 3): full GPU suite, smoke, the driver's bench line, rocprofv3 kernel trace + work log -> per-kernel roofline
# table, the full-pipeline line, the default 2-rank plan on one shared GPU.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
grep -E "^\[|passed|failed|FAILED" $O/gpu_tests_full.log | grep -v "Gloo\|W924\|c10d" > $O/r03_gpu_test_lines.txt; tail -1 $O/r03_gpu_test_lines.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke") > $O/smoke.txt; cat $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r03_bench_stage1_final.json 2> $O/bench.err; cut -c1-200 $O/r03_bench_stage1_final.json
cd /tmp && export TMPDIR=/tmp
SVD_WORKLOG=$O/wl.json timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_rt -o rt -- python $R/bench.py --workload ar_chunk --steps 2 --warmup 1 --no-trace --no-cpu-baseline > $O/rt_bench.log 2>&1
DB=$(find $O/prof_rt -name "*.db" | head -1)
python $R/tools/roofline_table.py $DB $O/wl.json $O/r03_roofline_table.txt > /dev/null 2>$O/rt_err.log
python $R/tools/rocprof_summary.py $DB $O/r03_ar_chunk_kernel_stats.txt > /dev/null 2>>$O/rt_err.log
rm -rf $O/prof_rt; head -6 $O/r03_roofline_table.txt | cut -c1-200
cd $R
timeout 700 python bench.py --workload full --steps 1 --warmup 0 > $O/r03_bench_full_pipeline.json 2>$O/bench_full.err; cut -c1-160 $O/r03_bench_full_pipeline.json
SVD_BENCH_SHARE_GPU=1 timeout 500 python bench.py --gpus 2 --denoise-steps 2 --steps 6 --warmup 0 --no-trace --no-cpu-baseline > $O/r03_bench_2rank_shared_gpu.json 2>$O/bench_2rank.err; cut -c1-160 $O/r03_bench_2rank_shared_gpu.json
timeout 300 python tools/host_bound_probe.py > $O/r03_host_probe.txt 2>&1; tail -3 $O/r03_host_probe.txt
