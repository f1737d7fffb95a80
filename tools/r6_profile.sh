#!/bin/bash
# Round-6 profile artefacts in one go (-> profiles/r06_*): driver-style bench line, the STEADY-STATE rocprofv3 --kernel-trace table
# per queue (tools/prof_streams.py --steady: only the last n optimizer steps, model construction and warm-up outside the window),
# the rocprofv3 --stats summary of the same command, the two --pmc traffic passes, the SQ busy-counter passes over the real step.
#     bash tools/r6_profile.sh <tag>
T=${1:-r6/final}
mkdir -p gpurun_out/$T/prof gpurun_out/$T/traffic gpurun_out/$T/pmc
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/$T/bench_final.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/prof -o r6 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-retry > $R/gpurun_out/$T/prof.log 2>&1
cd $R
DB=$(find gpurun_out/$T/prof -name "*.db" | head -1)
python tools/prof_streams.py $DB --steady 6 1 > gpurun_out/$T/q1.txt 2>&1
for q in 2 3 4; do python tools/prof_streams.py $DB --steady 6 $q >> gpurun_out/$T/qx.txt 2>&1; done
find gpurun_out/$T/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$T/kernel_stats.csv \;
rm -rf gpurun_out/$T/prof
timeout 900 bash tools/pmc_bench.sh $T/traffic > gpurun_out/$T/traffic.txt 2>&1
rm -rf gpurun_out/$T/traffic/FETCH_SIZE gpurun_out/$T/traffic/WRITE_SIZE
timeout 900 bash tools/pmc_step.sh $T/pmc > gpurun_out/$T/pmc.txt 2>&1
rm -rf gpurun_out/$T/pmc/s1 gpurun_out/$T/pmc/s2
timeout 300 python bench.py --workload videoqa --steps 10 --warmup 3 > gpurun_out/$T/bench_videoqa.json 2>/dev/null
timeout 300 python bench.py --workload mc --steps 10 --warmup 3 > gpurun_out/$T/bench_mc.json 2>/dev/null
head -30 gpurun_out/$T/q1.txt; tail -3 gpurun_out/$T/traffic.txt; head -20 gpurun_out/$T/pmc.txt; cut -c1-250 gpurun_out/$T/bench_final.json
