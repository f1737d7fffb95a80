#!/bin/bash
# round 4, call 4: full GPU suite + full bench line (extras: train loop, downstream lines)
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/c4_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c4_rc.txt
timeout 600 python bench.py > gpurun_out/r4/c4_bench.json 2> gpurun_out/r4/c4_bench.err; echo "bench rc=$?" >> gpurun_out/r4/c4_rc.txt
cat gpurun_out/r4/c4_rc.txt; tail -15 gpurun_out/r4/c4_pytest.log; tail -5 gpurun_out/r4/c4_bench.err; cut -c1-300 gpurun_out/r4/c4_bench.json
