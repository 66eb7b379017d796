#!/bin/bash
# Round-4 hardware counters (separate rocprofv3 --pmc passes, no trace domains): HBM traffic (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of the five
# heaviest plain-GEMM signatures -> <out>/r04_traffic_signatures.json; SQ occupancy / MFMA-busy counters of three of them and of the level-0
# spatial attention -> <out>/r04_pmc_sq_*.txt.      bash tools/pmc_round4.sh <out-dir>
set -u
O=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O
for SIG in m0_M460800_N2560_K320_s0_u0_e1_o0 m0_M115200_N5120_K640_s0_u0_e1_o0 m0_M28800_N10240_K1280_s0_u0_e1_o0 m0_M460800_N320_K1280_s0_u0_e0_o0 m0_M460800_N320_K320_s0_u0_e0_o0; do
  bash $R/tools/pmc_signature.sh $SIG $O/r04_traffic_signatures.json
done
bash $R/tools/pmc_sq.sh m0_M460800_N2560_K320_s0_u0_e1_o0 m0_M460800_N320_K1280_s0_u0_e0_o0 m0_M28800_N10240_K1280_s0_u0_e1_o0 > /dev/null
for f in $R/gpurun_out/pmc_sq_m0_*.txt; do cp $f $O/r04_$(basename $f); done
# spatial attention, level 0: traffic + SQ counters
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP -d $R/gpurun_out/pmc_attn_$i -o q -- python $R/tools/attn_sig_run.py 3 > /dev/null 2>&1
done
python - $(for j in 1 2 3 4; do find $R/gpurun_out/pmc_attn_$j -name "*.db" | head -1; done) > $O/r04_pmc_attn_spatial_50x9216x5.txt <<'PY'
import sqlite3, sys
print("# attn_spatial_d64_kernel, 50 frames x 9216 tokens x 5 heads, fp16, launched alone (per-launch means; FETCH_SIZE / WRITE_SIZE in KiB, FETCH x2 on gfx950 for coalesced reads;")
print("# SQ_*CYCLES / WAIT / ACTIVE count quad-cycles summed over waves, MFMA_BUSY cycles summed over SIMDs); algorithmic: 4 N^2 d = 5.436 TFLOP, q, k, v read + o written once = 1180 MB")
for db in sys.argv[1:]:
    try:
        rows = sqlite3.connect(db).execute("select counter_name, kernel_name, value from counters_collection").fetchall()
    except Exception as e:
        print("no counters in", db, e); continue
    acc = {}
    for c, k, v in rows:
        if "attn_spatial" in k:
            a = acc.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    for c, (n, s) in sorted(acc.items()):
        print(f"{c:34s} {s / n:16.1f}   ({n} launches)")
PY
rm -rf $R/gpurun_out/pmc_attn_* $R/gpurun_out/pmc_sq_m0_*.txt
cat $O/r04_traffic_signatures.json | head -40; cat $O/r04_pmc_attn_spatial_50x9216x5.txt
