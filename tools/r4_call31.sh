#!/bin/bash
# captured step keeps the side-stream operands of the position-table chains referenced: graph tests, graphed step time
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "graphed or tiny_backward or inference_graph" > $O/c31_tests.log 2>&1; echo "tests rc=$?" > $O/c31_rc.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --training-graphs > $O/c31_graphed.json 2>/dev/null
cat $O/c31_rc.txt; tail -1 $O/c31_tests.log
python -c "import json;d=json.loads(open('$O/c31_graphed.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])"
