#!/bin/bash
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd $R
K="xlarge_golden or bf16_operand_oracle_tight or xlarge_backward_golden or output_attentions or two_rank_data_parallel_at_xlarge or tiny_forward_golden"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_goldens.py -q -s -k "$K or g3b" > gpurun_out/r4/c11_prec_now.log 2>&1
(cd _r3 && timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "xlarge_golden or bf16_operand_oracle_tight or xlarge_backward_golden or tiny_forward_golden" > $R/gpurun_out/r4/c11_prec_r3.log 2>&1)
grep -E "max_abs|max err|argmax|worst|G3b|attention prob|passed|failed|backend" gpurun_out/r4/c11_prec_now.log | cut -c1-260
echo ==== r3; grep -E "max_abs|max err|argmax|worst|passed|failed" gpurun_out/r4/c11_prec_r3.log | cut -c1-260
