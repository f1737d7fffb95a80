#!/bin/bash
# round 4, call 6: fixes of call 5 (stage release by bucket, deterministic sumsq) + host issue time
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c6_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c6_rc.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_model.py -q -k "two_rank or delayed or graphed" > gpurun_out/r4/c6_pytest_rep$i.log 2>&1; echo "rep$i rc=$?" >> gpurun_out/r4/c6_rc.txt; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4/c6_bench.json 2> gpurun_out/r4/c6_bench.err; echo "bench rc=$?" >> gpurun_out/r4/c6_rc.txt
cat gpurun_out/r4/c6_rc.txt; tail -8 gpurun_out/r4/c6_pytest.log; tail -3 gpurun_out/r4/c6_bench.err; cut -c1-200 gpurun_out/r4/c6_bench.json
