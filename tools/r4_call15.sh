#!/bin/bash
# QKV projection as 224-row tiles vs the skewed launch; the graphed step under the profiler; train_one_epoch with graphs
O=gpurun_out/r4; mkdir -p $O/c15prof
R=$GRAFT_REPO_ROOT; cd $R
DBG=$R/frozenbilm_amd/libfbl_dbg.so
FBL_LIB=$DBG timeout 200 python tools/bench_gemm.py --iters 30 --set hot > $O/c15_gemm_default.txt 2>&1
FBL_LIB=$DBG FBL_GEMM_PREF224=1 timeout 200 python tools/bench_gemm.py --iters 30 --set hot --check > $O/c15_gemm_pref224.txt 2>&1
for i in 1 2; do
  FBL_LIB=$DBG timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c15_step_default_$i.json 2>/dev/null
  FBL_LIB=$DBG FBL_GEMM_PREF224=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/c15_step_pref224_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline --training-graphs > $O/c15_step_graphed.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/c15prof -o g -- python $R/bench.py --steps 3 --warmup 2 --training-graphs --no-cpu-baseline > $R/$O/c15_prof.log 2>&1
cd $R
python tools/prof_streams.py $O/c15prof/g_results.db 17 1 > $O/c15_graph_q1.txt 2>&1
for q in 2 3 4 5 6; do python tools/prof_streams.py $O/c15prof/g_results.db 17 $q >> $O/c15_graph_qx.txt 2>&1; done
python tools/prof_summary.py $O/c15prof/g_results.db 17 45 > $O/c15_graph_all.txt 2>&1
rm -rf $O/c15prof
timeout 600 python bench.py --no-cpu-baseline --no-traffic > $O/c15_bench_full.json 2>/dev/null
grep -h "9024\|M=  8512 N=  6144" $O/c15_gemm_default.txt $O/c15_gemm_pref224.txt
for f in $O/c15_step_*.json; do echo $f $(python -c "import json,sys;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'])"); done
head -14 $O/c15_graph_q1.txt; head -12 $O/c15_graph_qx.txt
python -c "import json;d=json.loads(open('$O/c15_bench_full.json').read().strip().splitlines()[-1]);print(json.dumps(d['train_one_epoch'])[:900]);print(d['graphed_step'])"
