#!/bin/bash
# round-3 GPU run C: split stream kernel -- per-signature A/B, retune (both stream forms), stage-1 bench A/B on one box, host probe, full pipeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
timeout 300 python tools/gemm_stream_ab.py > $O/gemm_stream_ab.txt 2>$O/ab.err
timeout 900 python tools/tune_gemm.py > $O/tune.log 2>$O/tune.err
cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles.json
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream fp32 > $O/bench_stream_fp32.json 2>/dev/null
timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream 16 > $O/bench_stream_16.json 2>/dev/null
timeout 400 python tools/host_bound_probe.py > $O/host_probe.txt 2>&1
timeout 700 python bench.py --workload full --steps 1 --warmup 0 > $O/bench_full.json 2>$O/bench_full.err
ls -la $O; cut -c1-330 $O/bench_stream_fp32.json $O/bench_stream_16.json
