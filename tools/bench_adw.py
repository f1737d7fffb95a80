"""HIP-event timing of the grouped adapter-gradient launch (fbl_adapter_bwd_dw) at the bench shape.

    PYTHONPATH=. python tools/bench_adw.py [adapters per launch = 16] [A = 192]

prints the time of one launch and per adapter; 16 adapters = 768 workgroups = one full round of the chip (three per CU).
Under rocprofv3 --pmc (tools/pmc_attn.sh style passes) the kernel name is adapter_dw_kernel.
"""
import sys

import torch

from frozenbilm_amd import lib as L

N, H = 8512, 1536
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
A = int(sys.argv[2]) if len(sys.argv) > 2 else 192
Ap = (A + 63) // 64 * 64
dev = "cuda"


def seg():
    z = torch.zeros(N, Ap, dtype=torch.bfloat16, device=dev)
    z[:, :A] = torch.relu(torch.randn(N, A, device=dev))
    dz = torch.zeros(N, Ap, dtype=torch.bfloat16, device=dev)
    dz[:, :A] = torch.randn(N, A, device=dev) * 0.1
    return torch.randn(N, H, device=dev).bfloat16(), z, dz, torch.randn(N, H, device=dev).bfloat16()


grp = [([seg()], torch.zeros(H, A, device=dev), torch.zeros(A, H, device=dev), torch.zeros(A, device=dev)) for _ in range(G)]
for _ in range(3):
    L.adapter_bwd_dw(grp, A=A)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.adapter_bwd_dw(grp, A=A)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1000
print(f"adapters {G}  A {A}: {t:.1f} us per launch, {t / G:.1f} us per adapter "
      f"({G * 2 * (N * H * 2 + N * Ap * 2) / t / 1e6:.2f} TB/s of operand bytes)")
