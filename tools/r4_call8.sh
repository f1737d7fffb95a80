#!/bin/bash
# round 4, call 8: same-box A/B of the attention / LayerNorm kernels: round-3 library, current, current without the seed word
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
( for i in 1 2; do
  echo "== r3 tree"; (cd _r3 && timeout 120 python tools/bench_attn.py)
  echo "== current"; timeout 120 python tools/bench_attn.py
  echo "== current, no seed word"; FBL_LIB=$R/frozenbilm_amd/libfbl_noseed.so timeout 120 python tools/bench_attn.py
  done
  echo "== rowops current"; timeout 120 python tools/bench_rowops.py
  echo "== rowops no seed word"; FBL_LIB=$R/frozenbilm_amd/libfbl_noseed.so timeout 120 python tools/bench_rowops.py
  echo "== bench r3"; (cd _r3 && timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200)
  echo "== bench current"; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200
  echo "== bench r3"; (cd _r3 && timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200)
  echo "== bench current"; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline | cut -c1-200
) > gpurun_out/r4/c8_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/c8_ab.txt
