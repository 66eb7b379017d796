#!/bin/bash
# One parametrised lease script (round 4; replaces the per-call r3*.sh files):  gpurun -- 'bash tools/gpu_session.sh <out-dir> <step> [<step> ...]'
# Every step writes under gpurun_out/<out-dir>/ and is bounded by its own timeout.  Steps:
#   ffprobe     tools/_bin/ff_fused_probe (round 5: fused feed-forward vs the two-launch path, same process)    i2vparity   tools/i2v_parity.py: enhancer UNet vs the vendored reference per precision plan    paritysigmas   A5 at three sigmas + the 2-step chunk golden
#   fftest      the fused feed-forward's kernel tests      fsptests   tests/test_gpu_fullsize_parity.py with durations      fixtests   the tests repaired after a suite run      bench6ff / bench6w   same-box bench A/B: SVD_FF_FUSED=1|0, SVD_FF_WAVES=8|4
#   pmcff       tools/pmc_ff.sh (counters of the fused feed-forward, both forms)      share2 / fullshare2 / fullshare4   bench.py with 2 / 4 ranks on the one leased GPU over gloo (SVD_BENCH_SHARE_GPU=1)
#   kernels     kernel-level GPU tests (GEMM tiles, norms)                norms       tools/norm_bench.py, packed / one-row LayerNorm and fused / separate GroupNorm finalize
#   tunear      tools/tune_gemm.py --only ar16 (A/B of kernel variants)      paritysweep   A5 at full size over the precision plans      bench6ab   bench6 with the plan off / default / without the UNet's part
#   tune        the full tuner (writes streamingt2v_amd/gemm_tiles.json, copied out)
#   spbisect    tools/sp_delta_bisect.py                                   bench6      bench.py --steps 6 --warmup 1 (no trace / CPU baseline)
#   bench       the driver's command (bench.py --steps 20 --warmup 5)      parity      tools/fullsize_parity.py (A5 at full size vs the reference golden)
#   profile     rocprofv3 kernel trace + work log of 2 AR chunks -> roofline table        suite       the whole GPU suite + smoke
#   enhance     bench.py --workload enhance + its rocprofv3 roofline table  full        bench.py --workload full
#   pmc         PMC counter passes of the dominant GEMM signatures and the spatial attention (tools/pmc_round4.sh)
#   phase       tools/gemm_phase_profile.py on a probe build (make -C streamingt2v_amd/csrc gvariant NAME=phase PROBE_DEFS=-DSVD_GEMM_PHASE_PROFILE PV_CFGS='...'); PHASE_CFGS / PHASE_SHAPES select
#   cfgab       tools/gemm_cfg_ab.py on a probe library with experimental tiles (CFGAB_LIB, CFGAB_CFGS): per-tile timings + bit checksums on the job's GEMM shapes
#   tailab / tailbench   kernel tests + tools/geglu_ab.py (/ bench6) of libsvdhip.so against a variant library libsvdhip_pv_tail0.so (any GEMM-source A/B: build the variant's GEMM objects with the switch, link with the main objects)
#   geluab(2)   A/B of the GEGLU epilogue against a variant library built with -DSVD_GEGLU_SCALAR_GELU (libsvdhip_pv_scalargelu.so): tests + tools/geglu_ab.py + bench6 (2: geglu_ab only)
#   round 6: rowproj (kernel tests + tools/rowproj_probe.py), bench6rp / enhrp (SVD_ROWPROJ=1|0 on the stage-1 job / the enhancer window), paneltest / bench6panel (W-panel tile walk:
#   GEMM tests, SVD_GEMM_PANEL=0 A/B), bench6stg (SVD_FF_STAGGER), bench6ln / fftest4 (the feed-forward's LayerNorm epilogue, both wave forms), retune (tile table re-tune + A/B), pmc6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for step in "$@"; do
  echo "== $step"; t0=$(date +%s)
  case $step in
    fftest)   timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -p no:cacheprovider -k "ff_geglu_fused" > $O/fftest.log 2>&1; grep -E "^\[ff fused" $O/fftest.log | awk 'NR%8==1' | head -12; tail -3 $O/fftest.log ;;
    fsptests) timeout 1200 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -x -s -p no:cacheprovider --durations=12 > $O/fsptests.log 2>&1; grep -E "^\[" $O/fsptests.log | cut -c1-250; tail -18 $O/fsptests.log ;;
    fixtests) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multiproc.py -m gpu -q -x -s -p no:cacheprovider -k "x3_convolution or enhancer_cfg_half_units" > $O/fixtests.log 2>&1; grep -E "enhancer on the job plan" $O/fixtests.log | cut -c1-260; tail -3 $O/fixtests.log ;;
    bench6w) for v in 8 4; do SVD_FF_WAVES=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_ffw$v.json 2>$O/bench6_ffw$v.err; echo "SVD_FF_WAVES=$v"; cut -c1-120 $O/bench6_ffw$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_ffw$v.json; tail -2 $O/bench6_ffw$v.err; done ;;
    paritysigmas1) timeout 900 python tools/fullsize_parity.py --dtype fp16 --which wrapper --cases all --chunk --timing > $O/fullsize_parity_sigmas.txt 2>$O/parity_sigmas.err; cat $O/fullsize_parity_sigmas.txt; tail -3 $O/parity_sigmas.err ;;
    bench6ff) for v in 1 0; do SVD_FF_FUSED=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_ff$v.json 2>$O/bench6_ff$v.err; echo "FF_FUSED=$v"; cut -c1-120 $O/bench6_ff$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_ff$v.json; tail -2 $O/bench6_ff$v.err; done ;;
    ffprobe)  timeout 400 tools/_bin/ff_fused_probe ${FFPROBE_M:-460800} 20 2>&1 | tee $O/ff_fused_probe.txt ;;
    i2vparity) timeout 900 python tools/i2v_parity.py --plans ${I2V_PLANS:-sweep} --fullres --timing > $O/i2v_parity.txt 2>$O/i2v_parity.err; cat $O/i2v_parity.txt; tail -3 $O/i2v_parity.err ;;
    paritysigmas) timeout 900 python tools/fullsize_parity.py --dtype fp16 --which wrapper --cases all --chunk --timing > $O/fullsize_parity_sigmas.txt 2>$O/parity_sigmas.err; cat $O/fullsize_parity_sigmas.txt; tail -3 $O/parity_sigmas.err
              timeout 600 python tools/fullsize_parity.py --dtype fp16 --which wrapper --cases all --plans level0io --timing > $O/fullsize_parity_sigmas_level0io.txt 2>$O/parity_sigmas_l0io.err; cat $O/fullsize_parity_sigmas_level0io.txt ;;
    kernels)  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -p no:cacheprovider -k "gemm_plain or many_tiles or geglu or implicit_views or groupnorm or layernorm or split3 or x3 or head" > $O/kernels.log 2>&1; grep -E "^\[(rows_split3|x3 conv|head)" $O/kernels.log | head -40; tail -3 $O/kernels.log ;;
    tunear)   timeout 300 python tools/tune_gemm.py --only ar16 --out $O/tiles_ar16.json > $O/tune_ar16.log 2>$O/tune_ar16.err; tail -3 $O/tune_ar16.log ;;
    newtests) timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_multiproc.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py -m gpu -q -x -s -p no:cacheprovider > $O/newtests.log 2>&1; grep -E "^\[" $O/newtests.log | grep -v Gloo | cut -c1-260 | head -40; tail -3 $O/newtests.log ;;
    paritylevel0io) timeout 900 python tools/fullsize_parity.py --dtype fp16 --which wrapper --plans level0io --timing > $O/fullsize_parity_level0io.txt 2>$O/parity_level0io.err; cat $O/fullsize_parity_level0io.txt ;;
    paritylevel0) timeout 900 python tools/fullsize_parity.py --dtype fp16 --which wrapper --plans level0 --timing > $O/fullsize_parity_level0.txt 2>$O/parity_level0.err; cat $O/fullsize_parity_level0.txt ;;
    paritysweep) timeout 900 python tools/fullsize_parity.py --dtype fp16 --which wrapper --plans sweep --timing > $O/fullsize_parity_sweep.txt 2>$O/parity_sweep.err; cat $O/fullsize_parity_sweep.txt; tail -3 $O/parity_sweep.err ;;
    attnpipe) for v in 0 1; do echo "SVD_ATTN_PIPE=$v"; SVD_ATTN_PIPE=$v timeout 200 python tools/attn_bench.py 2>/dev/null | tee $O/attn_bench_pipe$v.txt; done
              SVD_ATTN_PIPE=1 timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_product_sizes.py tests/test_gpu_i2v.py -m gpu -q -x -p no:cacheprovider -k "attn_spatial or attn_cross or attention" > $O/attn_pipe_tests.log 2>&1; tail -3 $O/attn_pipe_tests.log ;;
    share2)   SVD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload enhance --gpus 2 --denoise-steps 4 --steps 1 --warmup 0 --no-trace --no-cpu-baseline > $O/bench_enhance_2rank_shared_gpu.json 2>$O/share2_enh.err; cut -c1-700 $O/bench_enhance_2rank_shared_gpu.json; tail -2 $O/share2_enh.err
              SVD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --denoise-steps 2 --steps 6 --warmup 0 --no-trace --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2>$O/share2.err; cut -c1-300 $O/bench_2rank_shared_gpu.json; tail -2 $O/share2.err ;;
    bench16)  timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream 16 > $O/bench6_stream16_rim.json 2>/dev/null; cut -c50-75 $O/bench6_stream16_rim.json
              timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --residual-stream 16 --no-exact-rim > $O/bench6_all16.json 2>/dev/null; cut -c50-75 $O/bench6_all16.json
              timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_default.json 2>/dev/null; cut -c50-75 $O/bench6_default.json ;;
    fullshare4) SVD_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --workload full --gpus 4 --parallelism job --denoise-steps 2 --steps 1 --warmup 0 --no-trace --no-cpu-baseline > $O/bench_full_4rank_job_shared_gpu.json 2>$O/fullshare4.err; cut -c1-200 $O/bench_full_4rank_job_shared_gpu.json; grep -o '"parallelism": {[^}]*}' $O/bench_full_4rank_job_shared_gpu.json | cut -c1-900; tail -2 $O/fullshare4.err ;;
    fullshare2) SVD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --workload full --gpus 2 --denoise-steps 2 --steps 1 --warmup 0 --no-trace --no-cpu-baseline > $O/bench_full_2rank_shared_gpu.json 2>$O/fullshare2.err; cut -c1-200 $O/bench_full_2rank_shared_gpu.json; grep -o '"parallelism": {[^}]*}' $O/bench_full_2rank_shared_gpu.json | cut -c1-600; tail -2 $O/fullshare2.err ;;
    norms)    timeout 200 python tools/norm_bench.py > $O/norm_bandwidth_after.txt 2>$O/norm_after.err
              grep -h "layernorm\|gn_stats" $O/norm_bandwidth_after.txt | head -12 ;;
    tune)     timeout 400 python tools/tune_gemm.py > $O/tune.log 2>$O/tune.err; cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles.json; head -8 $O/tune.log ;;
    spbisect) timeout 400 python tools/sp_delta_bisect.py > $O/sp_delta_bisect.txt 2>$O/spbisect.err; tail -4 $O/sp_delta_bisect.txt; tail -3 $O/spbisect.err ;;
    phase)    SVD_LIB_FILE=libsvdhip_pv_phase.so timeout 300 python tools/gemm_phase_profile.py ${PHASE_CFGS:-20,8,17,21,22,18} ${PHASE_SHAPES:-} 2>&1 | grep -v amdgpu.ids | tee $O/gemm_phase_profile.txt ;;
    cfgab)    SVD_LIB_FILE=${CFGAB_LIB:-libsvdhip_pv_xpf.so} timeout 250 python tools/gemm_cfg_ab.py ${CFGAB_CFGS:-20,8,21,17,26,18,25,19,27} 20 2>&1 | grep -v amdgpu.ids | tee $O/gemm_cfg_ab.txt ;;
    tailab)   timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_plain or many_tiles or geglu or implicit_views or conv3x3 or temporal or x3" > $O/tail_tests.log 2>&1; tail -2 $O/tail_tests.log
              if grep -q failed $O/tail_tests.log; then echo "kernel tests FAILED: skipping the A/B"; else
              for L in libsvdhip_pv_tail0.so libsvdhip.so libsvdhip_pv_tail0.so libsvdhip.so; do echo "# lib $L" | tee -a $O/tail_ab.txt; SVD_LIB_FILE=$L timeout 120 python tools/geglu_ab.py 30 2>&1 | grep "M=" | tee -a $O/tail_ab.txt; done; fi ;;
    fftest4)  SVD_FF_WAVES=4 timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -p no:cacheprovider -k "ff_geglu_fused" > $O/fftest4.log 2>&1; grep -E "LayerNorm" $O/fftest4.log | awk 'NR%6==1' | head -8; tail -3 $O/fftest4.log ;;
    bench6ln) for v in 1 0 1 0; do SVD_FF_FUSED_LN=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_ffln$v.json 2>$O/bench6_ffln$v.err; echo "SVD_FF_FUSED_LN=$v"; cut -c1-120 $O/bench6_ffln$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_ffln$v.json; tail -2 $O/bench6_ffln$v.err; done ;;
    bench6stg) for v in 45 0 45 0; do SVD_FF_STAGGER=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_stg$v.json 2>$O/bench6_stg$v.err; echo "SVD_FF_STAGGER=$v"; cut -c1-120 $O/bench6_stg$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_stg$v.json; done
              for v in 0 45; do echo "SVD_FF_STAGGER=$v"; SVD_FF_STAGGER=$v timeout 100 python tools/ff_ab.py 2>&1 | grep -v "amdgpu.ids\|LayerNorm\|layernorm"; done > $O/ff_stagger_ab.txt; cat $O/ff_stagger_ab.txt ;;
    paneltest) timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" > $O/paneltest.log 2>&1; tail -3 $O/paneltest.log ;;
    bench6panel) for v in 1 0 1 0; do if [ $v = 0 ]; then export SVD_GEMM_PANEL=0; else unset SVD_GEMM_PANEL; fi; timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_panel$v.json 2>$O/bench6_panel$v.err; echo "panel walk=$v"; cut -c1-120 $O/bench6_panel$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_panel$v.json; done; unset SVD_GEMM_PANEL
              for v in 1 0; do if [ $v = 0 ]; then export SVD_GEMM_PANEL=0; else unset SVD_GEMM_PANEL; fi; timeout 300 python bench.py --workload enhance --steps 1 --warmup 1 --no-cpu-baseline --no-trace > $O/enh_panel$v.json 2>/dev/null; echo "enhance, panel walk=$v"; cut -c1-140 $O/enh_panel$v.json; done; unset SVD_GEMM_PANEL ;;
    rowproj)  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "rowproj" > $O/rowproj_test.log 2>&1; tail -3 $O/rowproj_test.log
              timeout 300 python tools/rowproj_probe.py 2>&1 | grep -v amdgpu.ids > $O/rowproj_probe.txt; cat $O/rowproj_probe.txt ;;
    bench6rp) for v in 1 0 1 0; do SVD_ROWPROJ=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_rp$v.json 2>$O/bench6_rp$v.err; echo "SVD_ROWPROJ=$v"; cut -c1-120 $O/bench6_rp$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_rp$v.json; tail -1 $O/bench6_rp$v.err | cut -c1-200; done ;;
    enhrp)    for v in 1 0 1 0; do SVD_ROWPROJ=$v timeout 300 python bench.py --workload enhance --steps 1 --warmup 1 --no-cpu-baseline --no-trace > $O/enh_rp$v.json 2>/dev/null; echo "enhance, SVD_ROWPROJ=$v"; cut -c1-140 $O/enh_rp$v.json; done ;;
    retune)   cp streamingt2v_amd/gemm_tiles.json streamingt2v_amd/gemm_tiles_prev.json; timeout 500 python tools/tune_gemm.py > $O/tune.log 2>$O/tune.err; cp streamingt2v_amd/gemm_tiles.json $O/gemm_tiles_retuned.json; head -5 $O/tune.log
              python - <<'PY'
import json
a = json.load(open("streamingt2v_amd/gemm_tiles_prev.json"))["table"]; b = json.load(open("streamingt2v_amd/gemm_tiles.json"))["table"]
ch = [(k, a[k]["cfg"], b[k]["cfg"]) for k in b if k in a and a[k]["cfg"] != b[k]["cfg"]]
print(len(b), "signatures,", len(ch), "changed,", len([k for k in b if k not in a]), "new")
for k, x, y in ch[:40]: print("  ", k, x, "->", y)
PY
              for v in gemm_tiles.json gemm_tiles_prev.json gemm_tiles.json gemm_tiles_prev.json; do SVD_GEMM_TILES=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_$v.out 2>/dev/null; echo "SVD_GEMM_TILES=$v"; cut -c1-120 $O/bench6_$v.out; done ;;
    tailsplit) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" > $O/tailtest.log 2>&1; tail -3 $O/tailtest.log
              for v in 1 0 1 0; do SVD_GEMM_TAIL=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace --no-stages > $O/bench6_tail$v.json 2>$O/bench6_tail$v.err; echo "SVD_GEMM_TAIL=$v"; cut -c1-120 $O/bench6_tail$v.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_tail$v.json; tail -1 $O/bench6_tail$v.err | cut -c1-200; done ;;
    svttests) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_f32.py -m gpu -q -x -p no:cacheprovider > $O/svttests.log 2>&1; tail -4 $O/svttests.log ;;
    tailbench) for L in libsvdhip_pv_tail0.so libsvdhip.so; do SVD_LIB_FILE=$L timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_$L.json 2>/dev/null; echo $L; cut -c50-75 $O/bench6_$L.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_$L.json; done ;;
    geluab2)  for L in libsvdhip_pv_scalargelu.so libsvdhip.so libsvdhip_pv_scalargelu.so libsvdhip.so; do SVD_LIB_FILE=$L timeout 120 python tools/geglu_ab.py 30 2>&1 | tee -a $O/geglu_ab.txt; done ;;
    geluab)   timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "geglu or many_tiles" > $O/geglu_tests.log 2>&1; tail -2 $O/geglu_tests.log
              for L in libsvdhip_pv_scalargelu.so libsvdhip.so libsvdhip_pv_scalargelu.so libsvdhip.so; do SVD_LIB_FILE=$L timeout 120 python tools/geglu_ab.py 30 2>/dev/null | tee -a $O/geglu_ab.txt; done
              for L in libsvdhip_pv_scalargelu.so libsvdhip.so; do SVD_LIB_FILE=$L timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_$L.json 2>/dev/null; echo $L; cut -c50-75 $O/bench6_$L.json; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $O/bench6_$L.json; done ;;
    bench6)   timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6.json 2>$O/bench6.err; cut -c1-420 $O/bench6.json ;;
    bench6ab) SVD_EXACT_RIM=0 SVD_CN_STREAM_F32=0 SVD_STREAM_F32_MIN_CH=0 timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_plan_off.json 2>$O/bench6_off.err
              timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_plan_default.json 2>$O/bench6_def.err
              SVD_STREAM_F32_MIN_CH=0 timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-trace > $O/bench6_plan_rim_cn.json 2>$O/bench6_rc.err
              for f in $O/bench6_plan_*.json; do echo $f; cut -c50-75 $f; grep -o '"chunk0_s_mean": [0-9.]*, "ar_chunk_s_mean": [0-9.]*' $f; done ;;
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
    suite)    timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=40 > $O/gpu_tests_full.log 2>&1
              grep -E "^\[|passed|failed|FAILED" $O/gpu_tests_full.log | grep -v "Gloo\|W924\|c10d" > $O/gpu_test_lines.txt; tail -1 $O/gpu_test_lines.txt; grep -A42 "slowest 40 durations" $O/gpu_tests_full.log > $O/gpu_test_durations.txt; head -14 $O/gpu_test_durations.txt
              (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke") > $O/smoke.txt; cat $O/smoke.txt ;;
    pmc)      bash tools/pmc_round4.sh $O ;;
    pmc6)     bash tools/pmc_round6.sh $O ;;
    i2vtests) timeout 900 python -m pytest tests/test_gpu_i2v.py tests/test_gpu_fullsize_parity.py -m gpu -q -x -s -p no:cacheprovider -k "i2v or enhancer or shipped_architecture" > $O/i2vtests.log 2>&1; grep -E "^\[|^\.\[" $O/i2vtests.log | cut -c1-250; tail -3 $O/i2vtests.log ;;
    share2job) SVD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --denoise-steps 2 --steps 6 --warmup 0 --no-trace --no-cpu-baseline --no-stages > $O/bench_2rank_shared_gpu.json 2>$O/share2.err; cut -c1-300 $O/bench_2rank_shared_gpu.json; grep -o '"parallelism": "[^"]*"' $O/bench_2rank_shared_gpu.json | cut -c1-400; tail -2 $O/share2.err
              SVD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --denoise-steps 2 --steps 6 --warmup 0 --no-trace --no-cpu-baseline --no-stages > $O/bench_4rank_job_shared_gpu.json 2>$O/share4.err; cut -c1-300 $O/bench_4rank_job_shared_gpu.json; grep -o '"parallelism": "[^"]*"' $O/bench_4rank_job_shared_gpu.json | cut -c1-400; tail -2 $O/share4.err ;;
    pmcff)    bash tools/pmc_ff.sh $O ;;
    *)        echo "unknown step $step" ;;
  esac
  echo "   ($step: $(( $(date +%s) - t0 )) s)"
done
