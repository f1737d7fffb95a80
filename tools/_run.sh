mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/r3/t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t10.log
{
python tools/bench_attn.py lpt
python tools/bench_attn.py lpt
} 2>&1 | grep -v amdgpu.ids | cut -c40-220 > gpurun_out/r3/a10.log
