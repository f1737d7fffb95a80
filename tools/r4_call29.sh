#!/bin/bash
# bench line after the extras were restricted to single-GPU runs; the N = 2 code path with extras (gloo, one GPU, 4 layers)
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 5 > $O/c29_bench.json 2> $O/c29_bench.err; echo "bench rc=$?" > $O/c29_rc.txt
FBL_BENCH_SHARE_GPU=1 timeout 280 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --layers 4 > $O/c29_n2.json 2> $O/c29_n2.err; echo "n2 rc=$?" >> $O/c29_rc.txt
cat $O/c29_rc.txt
python -c "
import json
d=json.loads(open('$O/c29_bench.json').read().strip().splitlines()[-1]); print(d['value'], sorted(k for k in d if isinstance(d[k],dict)))
d=json.loads(open('$O/c29_n2.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], sorted(k for k in d if isinstance(d[k],dict)))" 2>&1 | tail -3
tail -2 $O/c29_n2.err
