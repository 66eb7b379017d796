#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
for V in ring2 ring3; do SVD_LIB_FILE=libsvdhip_pv_$V.so timeout 300 python tools/gemm_stream_ab.py > $O/ab_$V.txt 2>$O/ab_$V.err; done
timeout 300 python tools/gemm_stream_ab.py > $O/ab_ring4.txt 2>$O/ab_ring4.err
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream fp32 > $O/bench_stream_fp32.json 2>/dev/null
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream 16 > $O/bench_stream_16.json 2>/dev/null
cut -c1-120 $O/ab_ring2.txt; cut -c1-120 $O/ab_ring3.txt; cut -c1-120 $O/ab_ring4.txt; cut -c1-140 $O/bench_stream_fp32.json $O/bench_stream_16.json
