#!/bin/bash
# HBM traffic of ONE GEMM signature (separate --pmc passes, no other trace domains): bash tools/pmc_signature.sh <signature> <out.json>
set -u
SIG=$1; OUT=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -o f -- python $R/tools/gemm_sig_run.py $SIG 5 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -o w -- python $R/tools/gemm_sig_run.py $SIG 5 > /dev/null 2>&1
python $R/tools/pmc_signature.py $SIG $(find $R/gpurun_out/pmc_f -name "*.db" | head -1) $(find $R/gpurun_out/pmc_w -name "*.db" | head -1) $OUT
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
