#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/c25_pytest.log 2>&1; echo "pytest rc=$?" > $O/c25_rc.txt
timeout 300 python bench.py --steps 10 --warmup 3 --packed-rows --no-cpu-baseline --no-roofline > $O/c25_packed.json 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $O/c25_padded.json 2>/dev/null
cat $O/c25_rc.txt; tail -2 $O/c25_pytest.log; grep -E "^FAILED|^E  " $O/c25_pytest.log | head
for f in packed padded; do python -c "import json;d=json.loads(open('$O/c25_$f.json').read().strip().splitlines()[-1]);print('$f', d['ms_per_step'], d['value'])"; done
