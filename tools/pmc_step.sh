#!/bin/bash
# SQ busy counters of every kernel of the real training step (north_star: "rocprof ... MFMA utilisation against peak"):
# two separate --pmc passes over bench.py (kernel-trace only, never combined with other trace domains), then
# tools/pmc_step_summary.py -> per-kernel table (profiles/r05_pmc.md).        bash tools/pmc_step.sh <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/$1/s$i -o p -- python $R/bench.py --steps 1 --warmup 0 --prewarm 2 \
    --no-cpu-baseline --no-roofline --no-extras --no-traffic --no-retry > $R/gpurun_out/$1/s$i.log 2>&1
done
cd $R; python tools/pmc_step_summary.py gpurun_out/$1 3
