#!/bin/bash
# round 4, call 7: tail microbench (tile shapes) + kernel stats + traffic passes of the state so far
mkdir -p gpurun_out/r4/prof gpurun_out/r4/traffic
R=$GRAFT_REPO_ROOT; cd $R
( timeout 120 python tools/bench_tail.py
  FBL_LIB=$R/frozenbilm_amd/libfbl_dbg.so FBL_GEMM_SMALL=1 timeout 120 python tools/bench_tail.py
  FBL_LIB=$R/frozenbilm_amd/libfbl_dbg.so FBL_GEMM_NO224=1 timeout 120 python tools/bench_tail.py ) > gpurun_out/r4/c7_tail.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4/prof -o r4 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/r4/c7_prof.log 2>&1
cd $R
python tools/prof_streams.py gpurun_out/r4/prof/r4_results.db 17 1 > gpurun_out/r4/c7_q1.txt 2>&1
python tools/prof_summary.py gpurun_out/r4/prof/r4_results.db 17 40 > gpurun_out/r4/c7_all.txt 2>&1
for q in 2 3 4; do python tools/prof_streams.py gpurun_out/r4/prof/r4_results.db 17 $q >> gpurun_out/r4/c7_qx.txt 2>&1; done
rm -rf gpurun_out/r4/prof
timeout 900 bash tools/pmc_bench.sh r4/traffic > gpurun_out/r4/c7_traffic.txt 2>&1
rm -rf gpurun_out/r4/traffic/FETCH_SIZE gpurun_out/r4/traffic/WRITE_SIZE
cat gpurun_out/r4/c7_tail.txt; head -30 gpurun_out/r4/c7_q1.txt; cat gpurun_out/r4/c7_traffic.txt | tail -20
