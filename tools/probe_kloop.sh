# K-loop timing probes (GPU box).  Variant libraries are built on the CPU side, one .so per variant, e.g.
#   make -C streamingt2v_amd/csrc pvariant NAME=pre PROBE_DEFS="-DSVD_GEMM_PROBES -DSVD_GEMM_PIECE_INTERLEAVE=0" PV_CFGS="$CFGS"
# and this script times them round-robin per shape (tools/gemm_probe2.py).   usage: bash tools/probe_kloop.sh "base new pre db1" 8,20,18,21,22,1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VARIANTS=${1:-"base new"}
CFGS=${2:-8}
for v in $VARIANTS; do
  echo "=== variant $v"
  SVD_LIB_FILE=libsvdhip_pv_$v.so timeout 400 python tools/gemm_probe2.py $CFGS 5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_gemm_kloop_probe2.log 2>&1
cat gpurun_out/r02_gemm_kloop_probe2.log
