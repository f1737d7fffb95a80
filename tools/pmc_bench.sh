#!/bin/bash
# HBM traffic of the GEMM family over the real step: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py,
# kernel-trace only (no sys/hip/hsa trace domains), then tools/pmc_bench_summary.py.   bash tools/pmc_bench.sh <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$1/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-retry > $R/gpurun_out/$1/$c.log 2>&1
done
cd $R; python tools/pmc_bench_summary.py gpurun_out/$1 15   # 12 pre-warm + 1 warm-up + 2 timed steps are instrumented
