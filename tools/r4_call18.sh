#!/bin/bash
# GEMM tile configurations at the row counts of packed batches
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
DBG=$R/frozenbilm_amd/libfbl_dbg.so
FBL_LIB=$DBG timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c18_default.txt 2>&1
FBL_LIB=$DBG FBL_GEMM_SMALL=1 timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c18_small.txt 2>&1
FBL_LIB=$DBG FBL_GEMM_NO224=1 timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c18_no224.txt 2>&1
FBL_LIB=$DBG FBL_GEMM8=0 timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c18_nogemm8.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_goldens.py -m gpu -q > $O/c18_goldens.log 2>&1
paste -d'|' <(cut -c1-62 $O/c18_default.txt) <(cut -c33-62 $O/c18_small.txt) <(cut -c33-62 $O/c18_no224.txt) <(cut -c33-62 $O/c18_nogemm8.txt)
tail -2 $O/c18_goldens.log
