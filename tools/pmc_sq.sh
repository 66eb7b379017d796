#!/bin/bash
# SQ occupancy / stall counters of ONE plain-GEMM signature launched alone (separate --pmc passes, no other trace domains):
#   bash tools/pmc_sq.sh <signature> [more signatures]   -> gpurun_out/pmc_sq_<sig>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for SIG in "$@"; do
  i=0
  for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
               "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $GROUP -d $R/gpurun_out/pmc_sq_$i -o q -- python $R/tools/gemm_sig_run.py $SIG 5 > /dev/null 2>&1
  done
  python - "$SIG" $(find $R/gpurun_out/pmc_sq_1 -name "*.db" | head -1) $(find $R/gpurun_out/pmc_sq_2 -name "*.db" | head -1) > $R/gpurun_out/pmc_sq_$SIG.txt <<'PY'
import sqlite3, sys
sig = sys.argv[1]
print("#", sig, "(per-launch means over the GEMM launches; SQ_*CYCLES / WAIT / ACTIVE count quad-cycles summed over waves, MFMA_BUSY cycles summed over SIMDs)")
for db in sys.argv[2:]:
    try:
        rows = sqlite3.connect(db).execute("select counter_name, kernel_name, value from counters_collection").fetchall()
    except Exception as e:
        print("no counters in", db, e); continue
    acc = {}
    for c, k, v in rows:
        if "gemm_kernel" in k:
            a = acc.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    for c, (n, s) in sorted(acc.items()):
        print(f"{c:34s} {s / n:16.1f}   ({n} launches)")
PY
  cat $R/gpurun_out/pmc_sq_$SIG.txt
  rm -rf $R/gpurun_out/pmc_sq_1 $R/gpurun_out/pmc_sq_2
done
