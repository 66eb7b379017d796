cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "1280 5120 8" "1280 5120 1" "2560 320 8" "320 2880 2"; do
  set -- $spec
  tag=N$1_K$2_c$3
  for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
    p=$(echo $pass | tr ' ' '_')
    timeout 120 rocprofv3 --pmc $pass --kernel-trace -d $R/gpurun_out/pmc_$tag_$p -o r -- python $R/tools/gemm_one.py $1 $2 $3 > /dev/null 2>&1
    db=$(find $R/gpurun_out/pmc_$tag_$p -name "*.db" | head -1)
    echo "== $tag $pass"; python $R/tools/pmc_summary.py $db gemm_kernel 2>&1 | tail -n +2
  done
done
