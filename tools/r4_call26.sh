#!/bin/bash
# in-situ time of the QKV projection as 224-row tiles (FBL_GEMM_PREF224=1) vs the skewed launch: kernel tables of both
O=gpurun_out/r4; mkdir -p $O/c26a $O/c26b
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export FBL_LIB=$R/frozenbilm_amd/libfbl_dbg.so
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/c26a -o g -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/$O/c26a.log 2>&1
FBL_GEMM_PREF224=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/c26b -o g -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/$O/c26b.log 2>&1
cd $R
python tools/prof_streams.py $O/c26a/g_results.db 17 1 > $O/c26a_q1.txt 2>&1
python tools/prof_streams.py $O/c26b/g_results.db 17 1 > $O/c26b_q1.txt 2>&1
rm -rf $O/c26a $O/c26b
head -12 $O/c26a_q1.txt | cut -c1-110; echo ---; head -12 $O/c26b_q1.txt | cut -c1-110
