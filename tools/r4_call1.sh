#!/bin/bash
# round 4, call 1: parity of the fused adapter tail + A/B bench + gemm8 epilogue decomposition
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/c1_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c1_rc.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c1_bench_tail.json 2> gpurun_out/r4/c1_bench_tail.err
FBL_NO_TAIL=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c1_bench_notail.json 2> gpurun_out/r4/c1_bench_notail.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c1_bench_tail2.json 2>/dev/null
FBL_NO_TAIL=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4/c1_bench_notail2.json 2>/dev/null
export FBL_LIB=$PWD/frozenbilm_amd/libfbl_dbg.so
for v in 3 7 11 19; do FBL_GEMM8_VAR=$v timeout 120 python tools/gemm8_epi_probe.py; done > gpurun_out/r4/c1_epi_probe.txt 2>&1
tail -3 gpurun_out/r4/c1_pytest.log; cat gpurun_out/r4/c1_epi_probe.txt
