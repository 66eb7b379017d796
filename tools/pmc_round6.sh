#!/bin/bash
# Round-6 hardware counters (separate rocprofv3 --pmc passes, no trace domains; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md HBM section):
# HBM traffic of the GEMM family's heaviest plain signatures -> <out>/r06_traffic_signatures.json, and of svd_rowgemm320 -> <out>/r06_pmc_rowgemm320.txt.
#     bash tools/pmc_round6.sh <out-dir>
set -u
O=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O
for SIG in m0_M115200_N5120_K640_s0_u0_e1_o0 m0_M28800_N10240_K1280_s0_u0_e1_o0 m0_M460800_N320_K320_s0_u0_e0_o1 m0_M115200_N640_K2560_s0_u0_e0_o1 m0_M460800_N960_K320_s0_u0_e0_o0; do
  bash $R/tools/pmc_signature.sh $SIG $O/r06_traffic_signatures.json
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_rg_f -o f -- python $R/tools/rowgemm_sig_run.py 460800 5 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_rg_w -o w -- python $R/tools/rowgemm_sig_run.py 460800 5 > /dev/null 2>&1
python - $(find $R/gpurun_out/pmc_rg_f -name "*.db" | head -1) $(find $R/gpurun_out/pmc_rg_w -name "*.db" | head -1) > $O/r06_pmc_rowgemm320.txt <<'PY'
import sqlite3, sys
def tot(db, c):
    rows = [v for n, v in sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (c,)).fetchall() if "rowgemm320" in n]
    return len(rows), sum(rows)
nf, f = tot(sys.argv[1], "FETCH_SIZE"); nw, w = tot(sys.argv[2], "WRITE_SIZE")
M = 460800
alg_r, alg_w = M * 320 * (2 + 4) + 204800, M * 320 * (4 + 2)
print("# svd_rowgemm320 (to_out form: X 16 bit + R fp32 in, Y fp32 + LayerNorm(Y) 16 bit out), M = 460 800, launched alone; per-launch means;")
print("# FETCH_SIZE x 2 (gfx950 correction for wide coalesced reads; the kernel's dword reads may be counted in full: both shown), WRITE_SIZE as is; KiB -> bytes")
print(f"launches {max(nf, nw)}")
print(f"read : FETCH_SIZE {f * 1024 / max(nf, 1) / 1e6:9.1f} MB  (x2: {2 * f * 1024 / max(nf, 1) / 1e6:9.1f} MB)   algorithmic {alg_r / 1e6:9.1f} MB")
print(f"write: WRITE_SIZE {w * 1024 / max(nw, 1) / 1e6:9.1f} MB                      algorithmic {alg_w / 1e6:9.1f} MB")
PY
rm -rf $R/gpurun_out/pmc_rg_f $R/gpurun_out/pmc_rg_w
cat $O/r06_traffic_signatures.json | head -60; cat $O/r06_pmc_rowgemm320.txt
