#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench_default.json 2>/dev/null
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream fp32 > $O/bench_stream_fp32.json 2>/dev/null
timeout 300 python tools/gemm_stream_ab.py > $O/ab.txt 2>/dev/null
(timeout 900 python -m pytest tests/test_gpu_ar_parity.py tests/test_gpu_stream_f32.py tests/test_gpu_fullsize_parity.py -q -s 2>&1 | grep -E "^\[|passed|failed|FAILED" | grep -v Gloo) > $O/tests.log 2>&1
cut -c1-140 $O/bench_default.json $O/bench_stream_fp32.json; tail -4 $O/tests.log | cut -c1-300
