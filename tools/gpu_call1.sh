set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/fullsize_parity.py --dtype both --which both > gpurun_out/r02_fullsize_parity.log 2>&1
tail -5 gpurun_out/r02_fullsize_parity.log
python tools/fullarch_parity.py --dtype both --which both > gpurun_out/r02_fullarch_parity.log 2>&1
tail -12 gpurun_out/r02_fullarch_parity.log
python -m pytest tests/test_gpu_ar_parity.py tests/test_gpu_kernels_product_sizes.py -m gpu -q -s -x > gpurun_out/r02_newtests.log 2>&1
grep -E "^\[|passed|failed|Error|error" gpurun_out/r02_newtests.log | tail -40
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_stage1_first.json 2> gpurun_out/r02_bench_stage1_first.err
tail -c 3000 gpurun_out/r02_bench_stage1_first.json; tail -5 gpurun_out/r02_bench_stage1_first.err
