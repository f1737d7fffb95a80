#!/bin/bash
# kernel stats + HBM-side traffic of one training step (profiles/r04_*): bash tools/r4_profile.sh <tag>
T=${1:-r4/final}
mkdir -p gpurun_out/$T/prof gpurun_out/$T/traffic
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $R/gpurun_out/$T/bench_before.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/prof -o r4 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$T/prof.log 2>&1
cd $R
python tools/prof_streams.py gpurun_out/$T/prof/r4_results.db 17 1 > gpurun_out/$T/q1.txt 2>&1
python tools/prof_summary.py gpurun_out/$T/prof/r4_results.db 17 45 > gpurun_out/$T/all.txt 2>&1
for q in 2 3 4; do python tools/prof_streams.py gpurun_out/$T/prof/r4_results.db 17 $q >> gpurun_out/$T/qx.txt 2>&1; done
rm -rf gpurun_out/$T/prof
timeout 900 bash tools/pmc_bench.sh $T/traffic > gpurun_out/$T/traffic.txt 2>&1
rm -rf gpurun_out/$T/traffic/FETCH_SIZE gpurun_out/$T/traffic/WRITE_SIZE
timeout 300 python bench.py --workload videoqa --steps 10 --warmup 3 > gpurun_out/$T/bench_videoqa.json 2>/dev/null
timeout 300 python bench.py --workload mc --steps 10 --warmup 3 > gpurun_out/$T/bench_mc.json 2>/dev/null
head -24 gpurun_out/$T/q1.txt; tail -3 gpurun_out/$T/traffic.txt; cut -c1-250 gpurun_out/$T/bench_before.json
