"""Times fbl_disent_attn_fwd at the bench shape (ablations via FBL_ATTN_DBG: 1 no gather, 2 no bias MFMAs, 4 no P.V, 8 no dropout)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
from frozenbilm_amd.model.relpos import rel_index_vector
B, S, nh, span2 = 32, 266, 24, 512
H = nh * 64; Sp = 320; dev = "cuda"
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.7).to(torch.bfloat16).to(dev)
pqk = (torch.randn(span2, 2 * H, generator=g) * 0.7).to(torch.bfloat16).to(dev)
tl = torch.randint(32, 257, (B,), generator=g); tl[-1] = 256
mask = torch.zeros(B, S, dtype=torch.int32)
for b in range(B): mask[b, :10 + tl[b]] = 1
mask = mask.to(dev)
klen = (mask * torch.arange(1, S + 1, device=dev, dtype=torch.int32)).amax(1).to(torch.int32).contiguous()
relidx = torch.from_numpy(rel_index_vector(S, 256, 512, 256).copy()).to(dev)
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, nh, S, device=dev)
for p in (0.1, 0.0):
    f = lambda: L.disent_attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], pqk[:, H:], pqk[:, :H], relidx, mask.view(-1), 1 / math.sqrt(192), ctx, lse, B, S, Sp, nh, span2, klen=klen, p_drop=p, seed=7)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"FBL_ATTN_DBG={os.environ.get('FBL_ATTN_DBG','0'):>2s} p_drop={p}: {s.elapsed_time(e)*50:.1f} us")
