"""Where the epilogue time of the 8-phase GEMM goes (debug library only): one process per FBL_GEMM8_VAR --
3 = shipped, 7 = main loop only, 11 = epilogue without its global stores, 19 = stores aliased onto 256 rows of C (the output
stays in L2: store issue without HBM write traffic).  Shapes: the FFN-up GEMM of the step with exactly three rounds of tiles
(M = 8192: no remainder launch) and as it runs (M = 8512), plain bf16 and GELU + GELU' (two bf16 outputs) epilogues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
L.load()
dev = "cuda"
print("# FBL_GEMM8_VAR=%s" % os.environ.get("FBL_GEMM8_VAR", "3"), flush=True)
for M, N, K in ((8192, 6144, 1536), (8512, 6144, 1536), (8192, 6144, 512), (8192, 6144, 6144)):
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.rand(N, device=dev)
    for var, kw in (("bf16", dict(out_bf16=o)), ("gelugrad", dict(out_bf16=o, out_pre=o2, act=L.ACT_GELU_GRAD))):
        for _ in range(3): L.gemm(A, B, bias=bias, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): L.gemm(A, B, bias=bias, **kw)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 50
        print(f"M={M} N={N} K={K:5d} {var:9s} {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF", flush=True)
