"""Run ONE gemm shape a few times (for rocprofv3 --pmc passes).  usage: python tools/pmc_gemm.py M N K [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
var = sys.argv[4] if len(sys.argv) > 4 else "bf16"
L.load()
A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
o16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
o32 = torch.empty(M, N, dtype=torch.float32, device="cuda") if var == "f32" else None
for _ in range(6):
    L.gemm(A, B, out_bf16=o16 if var != "f32" else None, out_f32=o32)
torch.cuda.synchronize()
