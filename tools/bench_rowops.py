"""HIP-event timings of the LayerNorm kernels at the bench shape (N = 8512, H = 1536, dropout 0.1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
L.load()
dev = "cuda"
N, H = 8512, 1536
print("# " + (os.environ.get("FBL_LIB") or "libfbl.so"), flush=True)
y, r = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
t = torch.empty(N, H, device=dev); st = torch.empty(N, 2, device=dev); ob = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
dout = torch.randn(N, H, device=dev); dt = torch.empty(N, H, device=dev); dyb = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
dg, db, dys = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
ws = L.ln_bwd_ws(H, dev)
def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
L.ln_fwd(y=y, p_drop=0.1, seed=5, r_plain=r, gamma=g, beta=b, eps=1e-7, out_t=t, out_stats=st, out_bf16=ob, N=N, H=H)
for rep in range(2):
    print(f"ln_fwd full   {timeit(lambda: L.ln_fwd(y=y, p_drop=0.1, seed=5, r_plain=r, gamma=g, beta=b, eps=1e-7, out_t=t, out_stats=st, out_bf16=ob, N=N, H=H)):7.1f} us")
    print(f"ln_fwd stats  {timeit(lambda: L.ln_fwd(y=t, gamma=g, beta=b, eps=1e-7, out_stats=st, out_bf16=ob, N=N, H=H)):7.1f} us")
    print(f"ln_bwd (+fold){timeit(lambda: L.ln_bwd(dout, t, st, g, p_drop=0.1, seed=5, out_dt=dt, out_dy_bf16=dyb, dgamma=dg, dbeta=db, dysum=dys, ws=ws)):7.1f} us")
    print(f"ln_bwd p=0    {timeit(lambda: L.ln_bwd(dout, t, st, g, p_drop=0.0, seed=0, out_dt=dt, out_dy_bf16=dyb, dgamma=dg, dbeta=db, dysum=dys, ws=ws)):7.1f} us   (no mask regeneration: what the hash costs)")
