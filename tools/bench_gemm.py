"""Micro-benchmark of fbl_gemm_bf16_nt on the hot-path shapes (HIP events on the launch stream).
usage: python tools/bench_gemm.py [--iters 20]"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); args = ap.parse_args()
dev = "cuda"
L.load()
SHAPES = [  # (M, N, K, variant)
    (4096, 4096, 4096, "bf16"), (8192, 8192, 8192, "bf16"),
    (8512, 1536, 1536, "f32+bf16"), (8512, 1536, 1536, "bf16"), (8512, 4608, 1536, "bf16"),
    (8512, 6144, 1536, "gelu"), (8512, 6144, 1536, "bf16"), (8512, 1536, 6144, "f32+bf16"), (8512, 1536, 6144, "bf16"),
    (8512, 6144, 1536, "dgelu"), (8512, 1536, 6144, "addf32"), (8512, 1536, 4608, "addf32"),
    (8512, 128100, 1536, "logits"), (8512, 192, 1536, "relu"), (8512, 1536, 192, "addf32"),
]
for M, N, K, var in SHAPES:
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    ldc = (N + 63) // 64 * 64
    o16 = torch.empty(M, ldc, dtype=torch.bfloat16, device=dev) if var != "logits" else None
    o32 = torch.empty(M, ldc, dtype=torch.float32, device=dev) if var in ("f32+bf16", "addf32", "logits") else None
    bias = torch.zeros(N, device=dev)
    kw = dict(bias=bias, N=N)
    if var in ("bf16", "relu"): kw.update(out_bf16=o16, act=L.ACT_RELU if var == "relu" else L.ACT_NONE)
    elif var == "f32+bf16": kw.update(out_f32=o32, out_bf16=o16)
    elif var == "logits": kw.update(out_f32=o32)
    elif var == "gelu": kw.update(out_bf16=o16, out_pre=torch.empty_like(o16), act=L.ACT_GELU)
    elif var == "dgelu": kw.update(out_bf16=o16, aux=torch.randn(M, ldc, device=dev).to(torch.bfloat16), aux_kind=L.AUX_MUL_DGELU_BF16); kw.pop("bias")
    elif var == "addf32": kw.update(out_f32=o32, aux=torch.randn(M, ldc, device=dev), aux_kind=L.AUX_ADD_F32); kw.pop("bias")
    for _ in range(3): L.gemm(A, B, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters): L.gemm(A, B, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / args.iters
    print(f"M={M:6d} N={N:6d} K={K:5d} {var:9s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    del A, B, o16, o32, kw
