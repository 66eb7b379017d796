#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
timeout 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench_old_table.json 2>/dev/null
cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles_old.json
timeout 200 python tools/tune_gemm.py > $O/tune.log 2>$O/tune.err
cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles_new.json
timeout 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench_new_table.json 2>/dev/null
cut -c1-150 $O/bench_old_table.json $O/bench_new_table.json; head -8 $O/tune.log
