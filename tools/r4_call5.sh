#!/bin/bash
# round 4, call 5: device seed word + graphed training step + the tests call 4 did not reach
mkdir -p gpurun_out/r4
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "graphed or delayed or hidden_parameter or output_attentions or two_rank or train_mode or tiny_backward" > gpurun_out/r4/c5_pytest_a.log 2>&1; echo "pytest_a rc=$?" > gpurun_out/r4/c5_rc.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/c5_rc.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4/c5_bench.json 2> gpurun_out/r4/c5_bench.err; echo "bench rc=$?" >> gpurun_out/r4/c5_rc.txt
cat gpurun_out/r4/c5_rc.txt; tail -30 gpurun_out/r4/c5_pytest_a.log; tail -8 gpurun_out/r4/c5_pytest.log; tail -5 gpurun_out/r4/c5_bench.err; cut -c1-300 gpurun_out/r4/c5_bench.json
