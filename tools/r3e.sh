#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
grep -E "^\[|passed|failed" $O/gpu_tests_full.log | grep -v "Gloo\|W924" > $O/gpu_tests.log
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench_default.json 2>/dev/null
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream fp32 > $O/bench_stream_fp32.json 2>/dev/null
tail -3 $O/gpu_tests.log; cut -c1-140 $O/bench_default.json $O/bench_stream_fp32.json
