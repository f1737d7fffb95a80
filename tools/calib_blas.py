"""Calibration only (never used by the product): what does the vendor GEMM reach on the step's big shapes?"""
import torch, time
dev = "cuda"
shapes = [(8512, 6144, 1536), (8512, 1536, 6144), (8512, 4608, 1536), (8512, 1536, 1536), (8192, 8192, 8192), (4096, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        c = a @ b.t()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"{M}x{N}x{K}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TF")
