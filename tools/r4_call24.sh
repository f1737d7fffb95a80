#!/bin/bash
# 128-row 8-phase tiles: correctness (--check) and time at packed row counts; off / auto / forced
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
DBG=$R/frozenbilm_amd/libfbl_dbg.so
FBL_LIB=$DBG FBL_GEMM_R128=0 timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c24_off.txt 2>&1
FBL_LIB=$DBG FBL_GEMM_R128=2 timeout 200 python tools/bench_gemm.py --iters 30 --set packed --check > $O/c24_force.txt 2>&1
FBL_LIB=$DBG timeout 200 python tools/bench_gemm.py --iters 30 --set packed > $O/c24_auto.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm" > $O/c24_gemm_tests.log 2>&1
paste -d'|' <(cut -c1-62 $O/c24_off.txt) <(cut -c33-100 $O/c24_force.txt) <(cut -c33-62 $O/c24_auto.txt)
tail -2 $O/c24_gemm_tests.log
