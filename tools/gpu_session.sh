#!/bin/bash
# One parametrised lease script (round 4; replaces the per-call r3*.sh files):  gpurun -- 'bash tools/gpu_session.sh <out-dir> <step> [<step> ...]'
# Every step writes under gpurun_out/<out-dir>/ and is bounded by its own timeout.  Steps:
#   kernels     kernel-level GPU tests (GEMM tiles, norms)                norms       tools/norm_bench.py, packed / one-row LayerNorm and fused / separate GroupNorm finalize
#   tune2wg     tools/tune_gemm.py --only ar16 for three start de-phasing spans of the two-per-CU tiles
#   tune        the full tuner (writes streamingt2v_amd/gemm_tiles.json, copied out)
#   spbisect    tools/sp_delta_bisect.py                                   bench6      bench.py --steps 6 --warmup 1 (no trace / CPU baseline)
#   bench       the driver's command (bench.py --steps 20 --warmup 5)      parity      tools/fullsize_parity.py (A5 at full size vs the reference golden)
#   profile     rocprofv3 kernel trace + work log of 2 AR chunks -> roofline table        suite       the whole GPU suite + smoke
#   enhance     bench.py --workload enhance + its rocprofv3 roofline table  full        bench.py --workload full
#   pmc         PMC counter passes of the dominant GEMM signatures and the spatial attention (tools/pmc_round4.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for step in "$@"; do
  echo "== $step"; t0=$(date +%s)
  case $step in
    kernels)  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_plain or many_tiles or geglu or implicit_views or groupnorm or layernorm" > $O/kernels.log 2>&1; tail -3 $O/kernels.log ;;
    norms)    SVD_LN_PACKED=0 SVD_GN_FUSED_FINALIZE=0 timeout 200 python tools/norm_bench.py > $O/norm_bandwidth_before.txt 2>$O/norm_before.err
              timeout 200 python tools/norm_bench.py > $O/norm_bandwidth_after.txt 2>$O/norm_after.err
              grep -h "layernorm\|gn_stats" $O/norm_bandwidth_before.txt | head -12; echo --; grep -h "layernorm\|gn_stats" $O/norm_bandwidth_after.txt | head -12 ;;
    tune2wg)  for sp in 0 8 16; do SVD_GEMM_DEPHASE_2WG=$sp timeout 300 python tools/tune_gemm.py --only ar16 --out $O/tiles_span$sp.json > $O/tune_span$sp.log 2>$O/tune_span$sp.err; tail -2 $O/tune_span$sp.log | head -1; done ;;
    tune)     timeout 400 python tools/tune_gemm.py > $O/tune.log 2>$O/tune.err; cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles.json; head -8 $O/tune.log ;;
    spbisect) timeout 400 python tools/sp_delta_bisect.py > $O/sp_delta_bisect.txt 2>$O/spbisect.err; tail -4 $O/sp_delta_bisect.txt; tail -3 $O/spbisect.err ;;
    bench6)   timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6.json 2>$O/bench6.err; cut -c1-420 $O/bench6.json ;;
    bench)    timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_stage1.json 2>$O/bench.err; cut -c1-420 $O/bench_stage1.json ;;
    parity)   timeout 600 python tools/fullsize_parity.py > $O/fullsize_parity.txt 2>$O/parity.err; tail -12 $O/fullsize_parity.txt ;;
    profile)  (cd /tmp && export TMPDIR=/tmp && SVD_WORKLOG=$O/wl.json timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_rt -o rt -- python $R/bench.py --workload ar_chunk --steps 2 --warmup 1 --no-trace --no-cpu-baseline > $O/rt_bench.log 2>&1)
              DB=$(find $O/prof_rt -name "*.db" | head -1)
              python tools/roofline_table.py $DB $O/wl.json $O/roofline_table.txt > /dev/null 2>$O/rt_err.log
              python tools/rocprof_summary.py $DB $O/ar_chunk_kernel_stats.txt > /dev/null 2>>$O/rt_err.log
              rm -rf $O/prof_rt; head -12 $O/roofline_table.txt | cut -c1-200 ;;
    enhance)  timeout 400 python bench.py --workload enhance --steps 1 --warmup 1 > $O/bench_enhance.json 2>$O/enhance.err; cut -c1-300 $O/bench_enhance.json
              (cd /tmp && export TMPDIR=/tmp && SVD_WORKLOG=$O/wl_enh.json timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_enh -o rt -- python $R/bench.py --workload enhance --steps 1 --warmup 0 --no-trace --no-cpu-baseline > $O/rt_enh.log 2>&1)
              DB=$(find $O/prof_enh -name "*.db" | head -1)
              python tools/roofline_table.py $DB $O/wl_enh.json $O/enhance_roofline_table.txt > /dev/null 2>$O/rt_enh_err.log
              python tools/rocprof_summary.py $DB $O/enhance_kernel_stats.txt > /dev/null 2>>$O/rt_enh_err.log
              rm -rf $O/prof_enh; head -12 $O/enhance_roofline_table.txt | cut -c1-200 ;;
    full)     timeout 900 python bench.py --workload full --steps 1 --warmup 0 > $O/bench_full_pipeline.json 2>$O/bench_full.err; cut -c1-400 $O/bench_full_pipeline.json ;;
    suite)    timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
              grep -E "^\[|passed|failed|FAILED" $O/gpu_tests_full.log | grep -v "Gloo\|W924\|c10d" > $O/gpu_test_lines.txt; tail -1 $O/gpu_test_lines.txt
              (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke") > $O/smoke.txt; cat $O/smoke.txt ;;
    pmc)      bash tools/pmc_round4.sh $O ;;
    *)        echo "unknown step $step" ;;
  esac
  echo "   ($step: $(( $(date +%s) - t0 )) s)"
done
