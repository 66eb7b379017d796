# K-loop timing probes (GPU box): builds are made with `make -C streamingt2v_amd/csrc pvariant NAME=<v> PROBE_DEFS=...` (see Makefile), one .so per variant; this script times them.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base nobar static stnb nods all3; do
  echo "=== variant $v"
  SVD_LIB_FILE=libsvdhip_pv_$v.so timeout 300 python tools/gemm_probe2.py 8 5 2>&1 | grep -E "ff1 geglu L0|ff2 L2|conv 1280|proj L0|Error|error"
done > gpurun_out/r02_gemm_kloop_probe.log 2>&1
cat gpurun_out/r02_gemm_kloop_probe.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --denoise-steps 2 --no-cpu-baseline --no-trace > $R/gpurun_out/r02_kt.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_kt -name "*.db" | head -1) $R/gpurun_out/r02_stage1_2steps_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/prof_kt
head -40 $R/gpurun_out/r02_stage1_2steps_kernel_stats.txt | cut -c1-190
