#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "packed" > $O/c21_packed.log 2>&1; echo "packed rc=$?" > $O/c21_rc.txt
cat $O/c21_rc.txt; grep -E "packed vs grid|worst relative|passed|failed|^E " $O/c21_packed.log | tail -14
