#!/bin/bash
# after removing the start-skew spin-wait: GEMM tests, hot shapes, the step
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q > $O/c27_kernels.log 2>&1; echo "kernels rc=$?" > $O/c27_rc.txt
timeout 200 python tools/bench_gemm.py --iters 30 --set hot > $O/c27_gemm.txt 2>&1
for i in 1 2 3; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c27_step_$i.json 2>/dev/null; done
cat $O/c27_rc.txt; tail -1 $O/c27_kernels.log; grep "9024\|N=  6144" $O/c27_gemm.txt
for i in 1 2 3; do python -c "import json;d=json.loads(open('$O/c27_step_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])"; done
