#!/bin/bash
# position-table gradient chains launched at the start of the NEXT attention backward (FBL_POS_WINDOW=1) vs right away
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
FBL_POS_WINDOW=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "tiny_backward or xlarge_backward or graphed" > $O/c30_tests.log 2>&1; echo "tests rc=$?" > $O/c30_rc.txt
for i in 1 2 3; do
  FBL_POS_WINDOW=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c30_off_$i.json 2>/dev/null
  FBL_POS_WINDOW=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c30_on_$i.json 2>/dev/null
done
cat $O/c30_rc.txt; tail -1 $O/c30_tests.log
for f in $O/c30_o*.json; do echo $f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],2), sorted(d['step_ms_gpu'])[5])"); done
