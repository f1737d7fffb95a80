#!/bin/bash
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r4/c13_pytest_s.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c13_rc.txt
cat gpurun_out/r4/c13_rc.txt; grep -E "passed|failed" gpurun_out/r4/c13_pytest_s.log | tail -2
