"""HIP-event timings of the attention kernels at the bench shape (B=32, S=266, nh=24, ragged masks as bench.py draws
them: the longest sample LAST in the batch).

    python tools/bench_attn.py [lpt|nat] [S] [B]

lpt = dispatch the samples longest first (`border`, what the engine does), nat = batch order.  With the debug library
(FBL_LIB=frozenbilm_amd/libfbl_dbg.so) FBL_ATTN_PLAINMAP=1 switches the XCD-aware workgroup mapping off and
FBL_ATTN_DBG ablates parts of the forward.
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from frozenbilm_amd import lib as L
from frozenbilm_amd.model.relpos import rel_index_vector

order = sys.argv[1] if len(sys.argv) > 1 else "lpt"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 266
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nh, span2 = 24, 512
H = nh * 64
Sp = (S + 63) // 64 * 64
dev = "cuda"
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.7).to(torch.bfloat16).to(dev)
pqk = (torch.randn(span2, 2 * H, generator=g) * 0.7).to(torch.bfloat16).to(dev)
T = 10
tl = torch.randint(32, S - T + 1, (B,), generator=g)
tl[-1] = S - T
if os.environ.get("FULL") == "1":  # every sample at full length: no empty workgroups, no load imbalance
    tl[:] = S - T
vl = torch.randint(1, T + 1, (B,), generator=g)
mask = torch.zeros(B, S, dtype=torch.int32)
for b in range(B):
    mask[b, : vl[b]] = 1
    mask[b, T: T + tl[b]] = 1
mask = mask.to(dev)
klen = (mask * torch.arange(1, S + 1, device=dev, dtype=torch.int32)).amax(1).to(torch.int32).contiguous()
border = torch.argsort(klen, descending=True, stable=True).to(torch.int32).contiguous() if order == "lpt" else None
relidx = torch.from_numpy(rel_index_vector(S, 256, 512, 256).copy()).to(dev)
q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
pq, pk = pqk[:, :H], pqk[:, H:]
scale = 1 / math.sqrt(192)
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, nh, S, device=dev)
dctx = torch.randn(B * S, H, device=dev).to(torch.bfloat16)
dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
Dv = torch.empty(B, nh, S, device=dev)
KT = torch.empty(nh, 64, B, Sp, dtype=torch.bfloat16, device=dev)
QT = torch.empty(nh, 64, B, Sp, dtype=torch.bfloat16, device=dev)
dS = torch.zeros(B, nh, Sp, Sp, dtype=torch.bfloat16, device=dev)
dST = torch.zeros(B, nh, Sp, Sp, dtype=torch.bfloat16, device=dev)
PKT = torch.empty(nh, 64, span2, dtype=torch.bfloat16, device=dev)
PQT = torch.empty(nh, 64, span2, dtype=torch.bfloat16, device=dev)
rv = rel_index_vector(S, 256, 512, 256)
rmin, rcnt = int(rv[0]), int(rv[-1]) - int(rv[0]) + 1
G1T = torch.empty(nh, B * (Sp // 32) * rcnt * 32, dtype=torch.bfloat16, device=dev)
G2T = torch.empty_like(G1T)
P = 0.1
LIN = int(os.environ.get('LIN', '128'))


def fwd():
    L.disent_attn_fwd(q, k, v, pk, pq, relidx, mask.view(-1), scale, ctx, lse, B, S, Sp, nh, span2, klen=klen, p_drop=P,
                      seed=7, border=border, lin=LIN)


psave = torch.zeros(B, nh, Sp, Sp, dtype=torch.bfloat16, device=dev)
msave = torch.zeros(B, nh, Sp // 64, S, device=dev)


def fwd_save():  # the training forward: also leaves the un-normalised probabilities for the backward
    L.disent_attn_fwd(q, k, v, pk, pq, relidx, mask.view(-1), scale, ctx, lse, B, S, Sp, nh, span2, klen=klen, p_drop=P,
                      seed=7, border=border, lin=LIN, psave=psave, msave=msave)


PQX = torch.empty(nh, 2 * Sp, 64, dtype=torch.bfloat16, device=dev)


def prep_x():  # the preparation of the fused route: K^T, PK^T (query-major shear pass), D, PQX
    L.attn_bwd_prep(q, k, pq, pk, dctx, ctx, None, KT, None, PKT, Dv, B, S, Sp, nh, span2, relidx=relidx, PQX=PQX)


PKX = torch.empty(nh, 2 * Sp, 64, dtype=torch.bfloat16, device=dev)


def prep_xx():  # the preparation when both halves run in Toeplitz form: D, PQX, PKX
    L.attn_bwd_prep(q, k, pq, pk, dctx, ctx, None, None, None, None, Dv, B, S, Sp, nh, span2, relidx=relidx, PQX=PQX, PKX=PKX)


def bwd_dq():
    L.disent_attn_bwd_dq(dS, k, PKX, dqkv[:, :H], B, S, Sp, nh, klen=klen, border=border)


def bwd_pk():  # kernel A from the saved probabilities + dK in place
    L.disent_attn_bwd_dspk(psave, msave, q, v, dctx, PQX, lse, Dv, scale, dqkv[:, H:2 * H], dqkv[:, 2 * H:], dS, dST, B, S, Sp, nh,
                           p_drop=P, seed=7, klen=klen, border=border)


def bwd_p():  # kernel A from the saved probabilities
    L.disent_attn_bwd_dsp(psave, msave, v, dctx, lse, Dv, scale, dqkv[:, 2 * H:], dS, dST, B, S, Sp, nh, p_drop=P, seed=7,
                          klen=klen, border=border)


from frozenbilm_amd.attn_bwd import _delta_ranges  # noqa: E402
import types  # noqa: E402

dlo, dcnt, dcmax = _delta_ranges(S, types.SimpleNamespace(position_buckets=256, max_rel=512, att_span=256), torch.device(dev))
EPG = int(os.environ.get("EPG", "1"))  # executions per fbl_attn_pos_grad launch (25 = the end-of-backward launch, cold operands)
dpos = torch.empty(EPG, nh, rcnt, 64, device=dev)
_xs1 = [dS] + [torch.empty_like(dS).copy_(dS) for _ in range(EPG - 1)]
_xs2 = [dST] + [torch.empty_like(dST).copy_(dST) for _ in range(EPG - 1)]
_ys = [qkv] + [qkv.clone() for _ in range(EPG - 1)]


def posgrad():
    L.attn_pos_grad(0, _xs1, [y[:, :H] for y in _ys], dlo, dcnt, dcmax, dpos, B, S, Sp, nh, rcnt, klen=klen)
    L.attn_pos_grad(1, _xs2, [y[:, H:2 * H] for y in _ys], dlo, dcnt, dcmax, dpos, B, S, Sp, nh, rcnt, klen=klen)


def shear0n():
    L.disent_attn_bwd_shear(0, dS, KT, PKT, relidx, dqkv[:, :H], None, B, S, Sp, nh, span2, klen=klen, rmin=rmin, rcnt=rcnt,
                            lin=128, border=border)


def shear1n():
    L.disent_attn_bwd_shear(1, dST, QT, PQT, relidx, dqkv[:, H:2 * H], None, B, S, Sp, nh, span2, klen=klen, rmin=rmin,
                            rcnt=rcnt, lin=128, border=border)


def prep():
    if os.environ.get("PREP_SPLIT") == "1":  # the five separate launches the fused preparation replaced
        L.attn_rowdot(dctx, ctx, Dv, B, S, nh)
        L.head_transpose(k, KT, B, S, Sp, nh, head_major=True)
        L.head_transpose(q, QT, B, S, Sp, nh, head_major=True)
        L.head_transpose(pk, PKT, 1, span2, span2, nh, head_major=False)
        L.head_transpose(pq, PQT, 1, span2, span2, nh, head_major=False)
    else:
        L.attn_bwd_prep(q, k, pq, pk, dctx, ctx, QT, KT, PQT, PKT, Dv, B, S, Sp, nh, span2)


def bwd_a():
    L.disent_attn_bwd_ds(q, k, v, dctx, pk, pq, relidx, mask.view(-1), lse, Dv, scale, dqkv[:, 2 * H:], dS, dST, B, S, Sp,
                         nh, span2, p_drop=P, seed=7, klen=klen, border=border, lin=LIN)


def shear0():
    L.disent_attn_bwd_shear(0, dS, KT, PKT, relidx, dqkv[:, :H], G1T, B, S, Sp, nh, span2, klen=klen, rmin=rmin, rcnt=rcnt,
                            lin=128, border=border)


def shear1():
    L.disent_attn_bwd_shear(1, dST, QT, PQT, relidx, dqkv[:, H:2 * H], G2T, B, S, Sp, nh, span2, klen=klen, rmin=rmin,
                            rcnt=rcnt, lin=128, border=border)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


side = torch.cuda.Stream()
ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()


def shear_both():  # the two shear passes are independent: second one on a side stream
    ev0.record()
    with torch.cuda.stream(side):
        side.wait_event(ev0)
        shear1()
        ev1.record()
    shear0()
    torch.cuda.current_stream().wait_event(ev1)


fwd_save()
prep()
prep_x()
prep_xx()
bwd_p()
res = {n: timeit(f) for n, f in (("fwd", fwd), ("fwd_save", fwd_save), ("prep", prep), ("prep_x", prep_x), ("bwd_a", bwd_a), ("bwd_p", bwd_p),
                                 ("bwd_pk", bwd_pk), ("prep_xx", prep_xx), ("bwd_dq", bwd_dq),
                                 ("shear0", shear0), ("shear1", shear1), ("shear0_nogt", shear0n), ("shear1_nogt", shear1n),
                                 (f"posgrad_x{EPG}", posgrad), ("shear0||1", shear_both))}
npairs = int(sum(((int(k) + 63) // 64) ** 2 for k in klen.tolist()) * nh)
tag = f"pairs={npairs} order={order} S={S} B={B} plainmap={os.environ.get('FBL_ATTN_PLAINMAP', '0')} dbg={os.environ.get('FBL_ATTN_DBG', '0')} occ={os.environ.get('FBL_ATTN_OCC', '-')} lin={LIN}"
print(tag + " | " + "  ".join(f"{n} {t:.1f}us" for n, t in res.items()) + f"  | bwd total (recompute) {res['prep'] + res['bwd_a'] + res['shear0'] + res['shear1']:.1f}us, (saved P) "
      f"{res['prep'] + res['bwd_p'] + res['shear0'] + res['shear1']:.1f}us, (fused dK) {res['prep_x'] + res['bwd_pk'] + res['shear0_nogt']:.1f}us, (+ Toeplitz dQ) {res['prep_xx'] + res['bwd_pk'] + res['bwd_dq']:.1f}us")
