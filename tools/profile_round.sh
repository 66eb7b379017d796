#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace of a short c2 bench + two PMC passes (HBM traffic) -> gpurun_out/.
# Usage (via gpurun): bash tools/profile_round.sh <tag>      then copy gpurun_out/<tag>_* into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --denoise-steps 2 --no-cpu-baseline --no-trace --warmup 0 --steps 1"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- $CMD > $R/gpurun_out/${TAG}_kt.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_kt -name "*.db" | head -1) $R/gpurun_out/${TAG}_c2_2steps_kernel_stats.txt > /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_f -o f -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_w -o w -- $CMD > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find $R/gpurun_out/prof_f -name "*.db" | head -1) $(find $R/gpurun_out/prof_w -name "*.db" | head -1) $R/gpurun_out/${TAG}_traffic_c2_2steps.json
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_f $R/gpurun_out/prof_w
head -30 $R/gpurun_out/${TAG}_c2_2steps_kernel_stats.txt
