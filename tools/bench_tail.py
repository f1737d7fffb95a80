"""HIP-event timing of the adapter tail (fbl_adapter_up_resid_fwd: up-projection + dropout + residual -> pre-norm tensor) and of
the LayerNorm statistics pass behind it at the bench shape; with the debug library FBL_GEMM_SMALL=1 runs the GEMM on 128x128
tiles (two workgroups per CU) instead of the 224x256 ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
L.load()
dev = "cuda"
M, H, A = 8512, 1536, 192
print("# " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FBL_")), flush=True)
z = torch.relu(torch.randn(M, A, device=dev)).to(torch.bfloat16)
Wu = (torch.randn(H, A, device=dev) * 0.05).to(torch.bfloat16)
bu = torch.randn(H, device=dev) * 0.1
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
rt = torch.randn(M, H, device=dev)
st = torch.stack([rt.mean(1), 1.0 / rt.std(1)], 1).contiguous()
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
t = torch.empty(M, H, device=dev)
stats = torch.empty(M, 2, device=dev); ob = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
tail = lambda: L.adapter_up_resid_fwd(z, Wu, bu, x, t, p_drop=0.1, seed=7, r_norm=(rt, st, g, b, None))
lnst = lambda: L.ln_fwd(y=t, gamma=g, beta=b, eps=1e-7, out_stats=stats, out_bf16=ob, N=M, H=H)
print(f"tail  {timeit(tail):7.1f} us   ({(M*H*(2+4+4)+M*A*2)/1e6:.0f} MB algorithmic)")
print(f"ln    {timeit(lnst):7.1f} us   ({M*H*6/1e6:.0f} MB)")
both = lambda: (tail(), lnst())
print(f"pair  {timeit(both):7.1f} us")
dy = torch.randn(M, H, device=dev).to(torch.bfloat16); upT = Wu.t().contiguous(); dz = torch.empty(M, A, dtype=torch.bfloat16, device=dev)
dzf = lambda: L.gemm(dy, upT, alpha=1.1, aux=z, aux_kind=L.AUX_MUL_POS_BF16, out_bf16=dz, N=A)
print(f"dz    {timeit(dzf):7.1f} us")
tail0 = lambda: L.adapter_up_resid_fwd(z, Wu, bu, x, t, p_drop=0.0, seed=0, r_norm=(rt, st, g, b, None))
print(f"tail p=0 {timeit(tail0):7.1f} us   (no dropout: what the hash costs)")
