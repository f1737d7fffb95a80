"""Attention forward + backward at the bench shape (B=32, S=266, nh=24, ragged masks) for rocprofv3 --pmc passes."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frozenbilm_amd import lib as L
from frozenbilm_amd.attn_bwd import disent_attn_bwd
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
border = torch.argsort(klen, descending=True, stable=True).to(torch.int32).contiguous()
relidx = torch.from_numpy(rel_index_vector(S, 256, 512, 256).copy()).to(dev)
class E: pass
eng, run, sv = E(), E(), E()
eng.H, eng.nh, eng.span2, eng.dev = H, nh, span2, torch.device(dev)
eng.relidx = lambda S_: relidx
import types as _t
eng.cfg = _t.SimpleNamespace(position_buckets=256, max_rel=512, att_span=256)
eng.sk_ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
P_ATT = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
run.B, run.S, run.mask_i32, run.p_att, run.klen, run.border = B, S, mask.view(-1), P_ATT, klen, border
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, nh, S, device=dev)
    L.disent_attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], pqk[:, H:], pqk[:, :H], relidx, mask.view(-1), 1 / math.sqrt(192), ctx, lse,
                      B, S, Sp, nh, span2, klen=klen, p_drop=P_ATT, seed=7, border=border, lin=128)
    sv.qkv, sv.pqk, sv.ctx, sv.lse, sv.seed_att = qkv, pqk, ctx, lse, 7
    dctx = torch.randn(B * S, H, device=dev).to(torch.bfloat16)
    dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
    dpqk = torch.zeros(span2, 2 * H, dtype=torch.bfloat16, device=dev)
    disent_attn_bwd(eng, run, sv, dctx, dqkv, dpqk)
torch.cuda.synchronize()
