#!/bin/bash
# isolated timings of the attention kernels at the bench shape: bash tools/prof_attn.sh <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1 -o a -- python $R/tools/pmc_attn.py 0.1 6 > /dev/null 2>&1
cd $R; python tools/prof_streams.py gpurun_out/$1/a_results.db 6 1 | head -14
