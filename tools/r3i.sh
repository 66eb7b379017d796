#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
(timeout 300 python -m pytest tests/test_gpu_graph.py -x -q 2>&1 | tail -15) > $O/graph_tests.log 2>&1; tail -5 $O/graph_tests.log
timeout 400 python tools/host_bound_probe.py > $O/host_probe.txt 2>&1; grep -v amdgpu $O/host_probe.txt | tail -8
timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --graph > $O/bench_graph.json 2>$O/bench_graph.err; cut -c1-150 $O/bench_graph.json; tail -2 $O/bench_graph.err
timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench_eager.json 2>/dev/null; cut -c1-150 $O/bench_eager.json
