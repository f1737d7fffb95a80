#!/bin/bash
# round 4, call 9: kernel A restored (two instantiations), host-side seed sums: full suite + A/B against the round-3 tree
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c9_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r4/c9_rc.txt
( echo "== attn current"; timeout 120 python tools/bench_attn.py
  echo "== attn r3"; (cd _r3 && timeout 120 python tools/bench_attn.py)
  for i in 1 2; do
  echo "== bench r3"; (cd _r3 && timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200)
  echo "== bench current"; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200
  done ) > gpurun_out/r4/c9_ab.txt 2>&1
cat gpurun_out/r4/c9_rc.txt; tail -5 gpurun_out/r4/c9_pytest.log; grep -v amdgpu.ids gpurun_out/r4/c9_ab.txt
