#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/c22_pytest.log 2>&1; echo "pytest rc=$?" > $O/c22_rc.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "packed_rows_training or graphed or delayed" >> $O/c22_repeat.log 2>&1; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c22_step.json 2>/dev/null
cat $O/c22_rc.txt; tail -2 $O/c22_pytest.log; grep -E "passed|failed" $O/c22_repeat.log; cut -c1-200 $O/c22_step.json
