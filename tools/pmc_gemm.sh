#!/bin/bash
# PMC passes over one GEMM shape (separate passes per counter group, kernel-trace only): bash tools/pmc_gemm.sh <tag> M N K
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/$1/p$i -o p -- python $R/tools/pmc_gemm.py $2 $3 $4 > /dev/null 2>&1
  python $R/tools/pmc_query.py $R/gpurun_out/$1/p$i/p_results.db gemm_bf16_nt
done
