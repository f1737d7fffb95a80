#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/c20_pytest.log 2>&1; echo "pytest rc=$?" > $O/c20_rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/c20_bench.json 2> $O/c20_bench.err; echo "bench rc=$?" >> $O/c20_rc.txt
cat $O/c20_rc.txt; tail -3 $O/c20_pytest.log; grep -E "^FAILED|^E " $O/c20_pytest.log | head
python -c "
import json;d=json.loads(open('$O/c20_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('packed_rows'))[:600]); print(json.dumps(d['videoqa_eval'].get('packed_rows')), d['videoqa_eval']['value']); print(json.dumps(d['mc_eval'].get('packed_rows')), d['mc_eval']['value'])" 2>&1 | tail -5
