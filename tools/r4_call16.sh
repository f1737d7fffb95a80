#!/bin/bash
# LayerNorm-backward fold with 32 lane groups x 8 loads in flight; shape sweep test
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/c16_pytest.log 2>&1; echo "pytest rc=$?" > $O/c16_rc.txt
timeout 200 python tools/bench_rowops.py > $O/c16_rowops.txt 2>&1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c16_step_$i.json 2>/dev/null
done
cat $O/c16_rc.txt; tail -3 $O/c16_pytest.log; cat $O/c16_rowops.txt | tail -12
for f in $O/c16_step_*.json; do echo $f $(python -c "import json,sys;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'])"); done
