#!/bin/bash
# Round-end validation on the GPU box (round 3): full GPU suite, smoke, the driver's bench line, rocprofv3 kernel trace + work log -> per-kernel roofline
# table, the full-pipeline line, the default 2-rank plan on one shared GPU.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
grep -E "^\[|passed|failed|FAILED" $O/gpu_tests_full.log | grep -v "Gloo\|W924\|c10d" > $O/r03_gpu_test_lines.txt; tail -1 $O/r03_gpu_test_lines.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke") > $O/smoke.txt; cat $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r03_bench_stage1_final.json 2> $O/bench.err; cut -c1-200 $O/r03_bench_stage1_final.json
timeout 400 python tools/host_bound_probe.py > $O/r03_host_probe.txt 2>&1; tail -3 $O/r03_host_probe.txt
