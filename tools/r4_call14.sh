#!/bin/bash
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c14_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c14_rc.txt
FBL_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --layers 4 > gpurun_out/r4/c14_n2.json 2> gpurun_out/r4/c14_n2.err; echo "n2 rc=$?" >> gpurun_out/r4/c14_rc.txt
FBL_FORCE_REDUCER=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r4/c14_forcered.json 2>/dev/null; echo "forcered rc=$?" >> gpurun_out/r4/c14_rc.txt
bash tools/r4_profile.sh r4/final2 > gpurun_out/r4/c14_profile.txt 2>&1
cat gpurun_out/r4/c14_rc.txt; grep -E "passed|failed" gpurun_out/r4/c14_pytest.log | tail -1; cut -c1-400 gpurun_out/r4/c14_n2.json; tail -3 gpurun_out/r4/c14_n2.err; cut -c1-200 gpurun_out/r4/c14_forcered.json; head -12 gpurun_out/r4/final2/q1.txt; tail -2 gpurun_out/r4/final2/traffic.txt
