#!/bin/bash
# Round-5 hardware counters of the fused feed-forward kernel (separate rocprofv3 --pmc passes, no trace domains):  bash tools/pmc_ff.sh <out-dir>
# -> <out>/r05_pmc_ff_fused_M460800.txt (four-wave default) and ..._8waves.txt (SVD_FF_WAVES=8)
set -u
O=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in 4 8; do
  i=0
  for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
               "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    SVD_FF_WAVES=$W timeout 200 rocprofv3 --pmc $GROUP -d $R/gpurun_out/pmc_ff_${W}_$i -o q -- python $R/tools/ff_sig_run.py 3 > /dev/null 2>&1
  done
  python - $W $(for j in 1 2 3 4; do find $R/gpurun_out/pmc_ff_${W}_$j -name "*.db" | head -1; done) > $O/r05_pmc_ff_fused_M460800_${W}waves.txt <<'PY'
import sqlite3, sys
print(f"# ff_geglu_fused kernel, {sys.argv[1]} waves per workgroup, M = 460 800 rows x C 320 x hidden 1280, fp16, fp32 residual in / fp32 out, launched alone (per-launch means; FETCH_SIZE / WRITE_SIZE in KiB,")
print("# FETCH x2 on gfx950 for coalesced reads; SQ_*CYCLES / WAIT / ACTIVE count quad-cycles summed over waves, MFMA_BUSY cycles summed over SIMDs); algorithmic: 6 M C H = 1.132 TFLOP;")
print("# X read (295 MB) + R read (590 MB) + Y written (590 MB) + the 2.4 MB weight image per workgroup tile from L2")
for db in sys.argv[2:]:
    try:
        rows = sqlite3.connect(db).execute("select counter_name, kernel_name, value from counters_collection").fetchall()
    except Exception as e:
        print("no counters in", db, e); continue
    acc = {}
    for c, k, v in rows:
        if "ff_geglu_fused" in k:
            a = acc.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    for c, (n, s) in sorted(acc.items()):
        print(f"{c:34s} {s / n:16.1f}   ({n} launches)")
PY
done
rm -rf $R/gpurun_out/pmc_ff_*
cat $O/r05_pmc_ff_fused_M460800_4waves.txt $O/r05_pmc_ff_fused_M460800_8waves.txt
