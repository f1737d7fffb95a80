#!/bin/bash
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_gpu_model.py -q -k "graphed" > gpurun_out/r4/c12_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c12_rc.txt
timeout 100 python __graft_entry__.py smoke > gpurun_out/r4/c12_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4/c12_rc.txt
timeout 900 python bench.py > gpurun_out/r4/c12_bench.json 2> gpurun_out/r4/c12_bench.err; echo "bench rc=$?" >> gpurun_out/r4/c12_rc.txt
cat gpurun_out/r4/c12_rc.txt; tail -2 gpurun_out/r4/c12_pytest.log; tail -2 gpurun_out/r4/c12_smoke.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4/c12_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:200])
print(d["roofline"].get("traffic_gemm8_gb_per_step"), d["roofline"].get("traffic_whole_step_gb"))
for k in ("eval_forward","graphed_step","host"):
    v=d.get(k); print(k, {a:b for a,b in v.items() if a in("value","ms_per_step","frac_of_peak","issue_ms_per_step")})
P
