"""Fixed-overhead probe of the 8-phase GEMM: time vs K at the step's N (plain bf16 epilogue).
Run once per FBL_GEMM8_VAR (3 = shipped, 7 = main loop only / nothing stored) and compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
L.load()
dev = "cuda"
print("# FBL_GEMM8_VAR=%s" % os.environ.get("FBL_GEMM8_VAR", "3"))
for N in (1536, 6144):
    for K in (256, 512, 1024, 1536, 3072, 6144):
        M = 8512
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for _ in range(3): L.gemm(A, B, out_bf16=o)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): L.gemm(A, B, out_bf16=o)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 50
        print(f"N={N:5d} K={K:5d} {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF", flush=True)
