#!/bin/bash
# round-3 GPU run B: new parity tests, roofline table profile, norm bandwidth, full-pipeline line, 2-rank self-launch on one GPU, host probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
(timeout 500 python -m pytest tests/test_gpu_ar_parity.py tests/test_gpu_multiproc.py -x -q -s 2>&1 | grep -E "^\[|passed|failed|Error|error" | tail -40) > $O/tests.log 2>&1
cd /tmp && export TMPDIR=/tmp
SVD_WORKLOG=$O/wl.json timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_rt -o rt -- python $R/bench.py --workload ar_chunk --steps 2 --warmup 1 --no-trace --no-cpu-baseline > $O/rt_bench.log 2>&1
DB=$(find $O/prof_rt -name "*.db" | head -1)
python $R/tools/roofline_table.py $DB $O/wl.json $O/r03_roofline_table.txt > /dev/null 2>$O/rt_err.log
python $R/tools/rocprof_summary.py $DB $O/r03_ar_chunk_kernel_stats.txt > /dev/null 2>>$O/rt_err.log
rm -rf $O/prof_rt
cd $R
timeout 200 python tools/norm_bench.py > $O/r03_norm_bandwidth.txt 2>$O/norm_err.log
timeout 600 python bench.py --workload full --steps 1 --warmup 0 > $O/bench_full.json 2>$O/bench_full.err
SVD_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 2 --denoise-steps 2 --steps 6 --warmup 0 --no-trace --no-cpu-baseline > $O/bench_2rank_shared.json 2>$O/bench_2rank_shared.err
timeout 300 python tools/host_bound_probe.py > $O/host_probe.txt 2>&1
ls -la $O
