#!/bin/bash
# kernel table of the packed step
O=gpurun_out/r4; mkdir -p $O/c19prof
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/c19prof -o g -- python $R/bench.py --steps 3 --warmup 2 --packed-rows --no-cpu-baseline --no-roofline > $R/$O/c19_prof.log 2>&1
cd $R
python tools/prof_streams.py $O/c19prof/g_results.db 17 1 > $O/c19_q1.txt 2>&1
for q in 2 3 4; do python tools/prof_streams.py $O/c19prof/g_results.db 17 $q >> $O/c19_qx.txt 2>&1; done
rm -rf $O/c19prof
timeout 300 python bench.py --steps 10 --warmup 3 --packed-rows --no-cpu-baseline --no-roofline > $O/c19_packed.json 2>/dev/null
head -40 $O/c19_q1.txt | cut -c1-120; head -8 $O/c19_qx.txt | cut -c1-120
python -c "import json;d=json.loads(open('$O/c19_packed.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_gpu'], d['step_ms_host'], d['host_loop_ms_per_step'])"
