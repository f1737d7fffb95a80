#!/bin/bash
# round 4, call 2: the two-workgroups-per-CU GEMM (gemm4.hip): parity + timing against the 8-phase kernel
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or dense_adapter" > gpurun_out/r4/c2_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c2_rc.txt
export FBL_LIB=$PWD/frozenbilm_amd/libfbl_dbg.so
( FBL_GEMM4=0 timeout 200 python tools/bench_gemm.py --iters 20 --check --set hot
  FBL_GEMM4=1 timeout 200 python tools/bench_gemm.py --iters 20 --check --set hot
  FBL_GEMM4=2 timeout 200 python tools/bench_gemm.py --iters 20 --check --set hot
  FBL_GEMM4=2 FBL_GEMM4_DBG=4 timeout 200 python tools/bench_gemm.py --iters 20 --set square
  FBL_GEMM4=0 timeout 200 python tools/bench_gemm.py --iters 20 --set square
  FBL_GEMM_SMALL=1 timeout 200 python tools/bench_gemm.py --iters 20 --set hot ) > gpurun_out/r4/c2_gemm.txt 2>&1
unset FBL_LIB
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c2_bench_g4.json 2> gpurun_out/r4/c2_bench_g4.err
cat gpurun_out/r4/c2_rc.txt; tail -3 gpurun_out/r4/c2_pytest.log; cat gpurun_out/r4/c2_gemm.txt; tail -c 1500 gpurun_out/r4/c2_bench_g4.json
