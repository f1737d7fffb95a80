#!/bin/bash
# packed rows: tests, then the whole suite, then the bench line with the packed_rows object
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "packed" > $O/c17_packed.log 2>&1; echo "packed rc=$?" > $O/c17_rc.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/c17_pytest.log 2>&1; echo "pytest rc=$?" >> $O/c17_rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/c17_bench.json 2> $O/c17_bench.err; echo "bench rc=$?" >> $O/c17_rc.txt
cat $O/c17_rc.txt; grep -E "packed vs grid|passed|failed|Error|error" $O/c17_packed.log | tail -12; tail -3 $O/c17_pytest.log
python -c "
import json;d=json.loads(open('$O/c17_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('packed_rows'))[:900])" 2>&1 | tail -3; tail -3 $O/c17_bench.err
