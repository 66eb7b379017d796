#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
SVD_ATTN_TPB=1 timeout 200 python tools/attn_bench.py > $O/attn_tpb1.txt 2>/dev/null
timeout 200 python tools/attn_bench.py > $O/attn_tpb2.txt 2>/dev/null
SVD_ATTN_TPB=1 timeout 200 python tools/attn_bench.py >> $O/attn_tpb1.txt 2>/dev/null
timeout 200 python tools/attn_bench.py >> $O/attn_tpb2.txt 2>/dev/null
(timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_product_sizes.py tests/test_gpu_i2v.py -q -k "attn or attention" 2>&1 | tail -3) > $O/attn_tests.log
SVD_LIB_FILE=libsvdhip_pv_ring3.so timeout 300 python tools/gemm_stream_ab.py > $O/ab_base_ring3.txt 2>/dev/null
timeout 300 python tools/gemm_stream_ab.py > $O/ab_main.txt 2>/dev/null
echo tpb1; cat $O/attn_tpb1.txt; echo tpb2; cat $O/attn_tpb2.txt; cat $O/attn_tests.log; echo ring3; cut -c1-110 $O/ab_base_ring3.txt | grep "16 bit"; echo main; cut -c1-110 $O/ab_main.txt | grep "16 bit"
