#!/bin/bash
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c10_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c10_rc.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4/c10_bench.json 2> gpurun_out/r4/c10_bench.err; echo "bench rc=$?" >> gpurun_out/r4/c10_rc.txt
cat gpurun_out/r4/c10_rc.txt; tail -6 gpurun_out/r4/c10_pytest.log; tail -3 gpurun_out/r4/c10_bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4/c10_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("eval_forward","videoqa_eval","mc_eval","graphed_step","host"):
    v=d.get(k); print(k, {a:b for a,b in v.items() if a in("value","ms_per_step","frac_of_peak","executed_frac_of_peak","issue_ms_per_step","eager_launches")})
P
