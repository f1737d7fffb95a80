#!/bin/bash
# round 4, call 3: branch-free interior epilogue: parity + timing
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x > gpurun_out/r4/c3_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c3_rc.txt
( timeout 200 python tools/bench_gemm.py --iters 20 --check --set hot
  export FBL_LIB=$PWD/frozenbilm_amd/libfbl_dbg.so
  for v in 3 7 11 19; do FBL_GEMM8_VAR=$v timeout 120 python tools/gemm8_epi_probe.py; done ) > gpurun_out/r4/c3_gemm.txt 2>&1
unset FBL_LIB
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c3_bench.json 2> gpurun_out/r4/c3_bench.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c3_bench2.json 2>/dev/null
cat gpurun_out/r4/c3_rc.txt; tail -3 gpurun_out/r4/c3_pytest.log; cat gpurun_out/r4/c3_gemm.txt; cut -c1-400 gpurun_out/r4/c3_bench.json; cut -c1-300 gpurun_out/r4/c3_bench2.json
