#!/bin/bash
# final verification of the round: full GPU suite, smoke(), then the profile artefacts
O=gpurun_out/r4; mkdir -p $O
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/c28_pytest.log 2>&1; echo "pytest rc=$?" > $O/c28_rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/c28_smoke.log 2>&1; echo "smoke rc=$?" >> $O/c28_rc.txt
bash tools/r4_profile.sh r4/final4 > $O/c28_profile.txt 2>&1
cat $O/c28_rc.txt; grep -E "passed|failed" $O/c28_pytest.log | tail -1; tail -2 $O/c28_smoke.log; head -8 $O/final4/q1.txt; tail -1 $O/final4/traffic.txt; cut -c1-220 $O/final4/bench_before.json
