cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02_tests5.log 2>&1; tail -4 gpurun_out/r02_tests5.log; grep -E "^\[SpatialVideoTransformer \+ APM|^\[sequence-parallel" gpurun_out/r02_tests5.log
timeout 900 python tools/tune_gemm.py > gpurun_out/r02_tune2.log 2>&1; grep -E "signatures|GEMM time" gpurun_out/r02_tune2.log; python - <<'PY'
import json,collections
t=json.load(open('streamingt2v_amd/gemm_tiles.json'))['table']
print(sorted(collections.Counter(int(v['cfg']) for v in t.values()).items()))
PY
timeout 900 python bench.py --steps 7 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg21.json 2> gpurun_out/r02_bench_cfg21.err; head -c 300 gpurun_out/r02_bench_cfg21.json; echo; tail -3 gpurun_out/r02_bench_cfg21.err
