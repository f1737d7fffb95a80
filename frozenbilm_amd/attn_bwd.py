"""Host-side orchestration of the disentangled-attention backward (see csrc/attn_bwd.hip for the math).

    prep     : D = rowdot(dO, O) + the position tables expanded by the index map (PQX / PKX)                       (one launch)
    dspk     : dV, dK, dS, dS^T from the probabilities the training forward saved (fbl_disent_attn_bwd_dspk)
    dq       : dQ = dS.K + the c2p term in Toeplitz form (fbl_disent_attn_bwd_dq)
    pos_grad : dPK[h] = sum_b G1^T.Q , dPQ[h] = sum_b G2^T.K -- straight from dS / dS^T, all layer executions at the end of
               backward (fbl_attn_pos_grad: the sheared operand G is formed on the fly out of LDS)
Earlier routes behind engine options (A/B measurements; calls without saved probabilities): attn_save_p = False -> kernel A
recomputes the probabilities (fbl_disent_attn_bwd_ds); attn_fused_dk / attn_toeplitz_dq = False -> the scatter-based shear
passes of rounds 1-5 for dK / dQ (with K^T, Q^T, PK^T, PQ^T from the preparation kernel); pos_grad_gt = True -> the shear passes
write G^T and two split-K GEMMs per table contract it against Q^T / K^T.
"""
from __future__ import annotations

import math

import torch

from . import lib as L

BF16, F32 = torch.bfloat16, torch.float32


POISON_GT = False
_RANGE = {}


def _relidx_range(S, cfg):
    """(first, count) of the position-table rows the relative-index map of a length-S sequence can touch (host integers,
    computed once per sequence length)"""
    key = (S, cfg.position_buckets, cfg.max_rel, cfg.att_span)
    if key not in _RANGE:
        from .model.relpos import rel_index_vector

        rv = rel_index_vector(S, cfg.position_buckets, cfg.max_rel, cfg.att_span)
        _RANGE[key] = (int(rv[0]), int(rv[-1]) - int(rv[0]) + 1)
    return _RANGE[key]


_DRANGE = {}


def _delta_ranges(S, cfg, dev):
    """(dlo, dcnt, max dcnt): int16 device tensors [rcnt]: table row rmin + r collects the deltas [dlo[r], dlo[r] + dcnt[r]) -- the inverse
    of the (monotone) relative-index vector, computed once per sequence length"""
    key = (S, cfg.position_buckets, cfg.max_rel, cfg.att_span, str(dev))
    if key not in _DRANGE:
        import numpy as np

        from .model.relpos import rel_index_vector

        rv = np.asarray(rel_index_vector(S, cfg.position_buckets, cfg.max_rel, cfg.att_span), dtype=np.int64)
        rmin, rcnt = int(rv[0]), int(rv[-1]) - int(rv[0]) + 1
        first = np.searchsorted(rv, np.arange(rmin, rmin + rcnt), side="left")
        last = np.searchsorted(rv, np.arange(rmin, rmin + rcnt), side="right")
        dlo = (first - (S - 1)).astype(np.int16)
        dcnt = (last - first).astype(np.int16)
        _DRANGE[key] = (torch.from_numpy(dlo).to(dev), torch.from_numpy(dcnt).to(dev), int(dcnt.max()))
    return _DRANGE[key]


def gt_tilemasks(eng, run):
    """(mask of G1^T, mask of G2^T) for this pass -- a function of the lengths and the relative-index map only, so one pair
    serves every layer execution: the shear passes skip the zero fill of the G^T rows outside the marked 128-row tiles and the
    position-table products skip fetching them (include/fbl.h fbl_gt_tilemask)."""
    m = getattr(run, "_gt_masks", None)
    if m is None:
        B, S = run.B, run.S
        Sp = (S + 63) // 64 * 64
        rmin, rcnt = _relidx_range(S, eng.cfg)
        klen = getattr(run, "klen", None)
        m = tuple(L.gt_tilemask(eng.relidx(S), klen, B, S, Sp, eng.span2, neg, rmin, rcnt) for neg in (0, 1))
        try:
            run._gt_masks = m
        except AttributeError:
            pass
    return m


def disent_attn_bwd(eng, run, sv, dctx, dqkv, dpqk, defer_pos=False, bufs=None):
    """defer_pos=True: skip the position-table GEMMs and return the state they need (the engine runs them for ALL layer
    executions at once at the end of backward: pos_table_grads_batched); otherwise dpqk [span2, 2H] (bf16, [dPQ|dPK]) is filled
    here.  bufs: optional pre-allocated (G1T, G2T, QT, KT) -- one execution's slices of the engine's per-step tensors."""
    B, S, H, nh, span2 = run.B, run.S, eng.H, eng.nh, eng.span2
    Sp = (S + 63) // 64 * 64
    dev = eng.dev
    qkv = sv.qkv
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    pq, pk = sv.pqk[:, :H], sv.pqk[:, H:]
    relidx = eng.relidx(S)
    klen = getattr(run, "klen", None)
    border = getattr(run, "border", None)
    scale = 1.0 / math.sqrt(64 * 3)
    pk_ = getattr(run, "pk", None)
    row0 = pk_.row0 if pk_ is not None else None  # packed-row layout of q / k / v / dO and of the dQ / dK / dV outputs

    # One launch prepares the backward: K^T, Q^T (head-major), PK^T, PQ^T and D_i = dO_i . O_i.  (Folding D into kernel A
    # was measured: +43 us there for the O tiles on its critical path; five separate small launches: 65 us in situ.)
    Dv = torch.empty(B, nh, S, dtype=F32, device=dev)
    use_gt = bool(getattr(eng, "pos_grad_gt", False))  # rounds 1-5: G^T through HBM + split-K GEMMs
    # round 6: with the forward's probabilities saved, kernel A forms dK itself (fbl_disent_attn_bwd_dspk) -- no key-major shear
    # pass, no Q^T / PQ^T copies; engine_options["attn_fused_dk"] = False keeps the separate pass
    fused_dk = getattr(sv, "psave", None) is not None and not use_gt and bool(getattr(eng, "attn_fused_dk", True))
    # the query-major half in Toeplitz form (fbl_disent_attn_bwd_dq) instead of the scatter-based shear pass: no K^T / PK^T copies;
    # engine_options["attn_toeplitz_dq"] = False keeps the shear pass
    toep_dq = not use_gt and bool(getattr(eng, "attn_toeplitz_dq", True))
    G1T = G2T = None
    QT = KT = PQT = PKT = PQX = PKX = None
    if bufs is not None and use_gt:
        G1T, G2T, QT, KT = bufs
    else:
        if not toep_dq:
            KT = torch.empty(nh, 64, B, Sp, dtype=BF16, device=dev)
        if not fused_dk:
            QT = torch.empty(nh, 64, B, Sp, dtype=BF16, device=dev)
    if toep_dq:
        PKX = torch.empty(nh, 2 * Sp, 64, dtype=BF16, device=dev)
    else:
        PKT = torch.empty(nh, 64, span2, dtype=BF16, device=dev)
    if fused_dk:
        PQX = torch.empty(nh, 2 * Sp, 64, dtype=BF16, device=dev)
    else:
        PQT = torch.empty(nh, 64, span2, dtype=BF16, device=dev)
    L.attn_bwd_prep(q, k, pq, pk, dctx, sv.ctx, QT, KT, PQT, PKT, Dv, B, S, Sp, nh, span2, row0=row0, relidx=relidx, PQX=PQX, PKX=PKX)
    dS = torch.empty(B, nh, Sp, Sp, dtype=BF16, device=dev)
    dST = torch.empty(B, nh, Sp, Sp, dtype=BF16, device=dev)
    # only the rows of G^T inside the range of relidx can be non-zero: write / contract just those
    rmin, rcnt = _relidx_range(S, eng.cfg)
    # |i-j| < lin: identity buckets, relidx injective (model/deberta.py:578-589: mid = bucket_size // 2)
    lin = eng.cfg.position_buckets // 2 if eng.cfg.position_buckets > 0 else 1 << 30
    lin_a = min(lin, span2 // 2) if eng.cfg.position_buckets > 0 else 0  # affine addressing of kernel A: identity buckets only
    if fused_dk:
        L.disent_attn_bwd_dspk(sv.psave, sv.msave, q, v, dctx, PQX, sv.lse, Dv, scale, dqkv[:, H:2 * H], dqkv[:, 2 * H:], dS, dST,
                               B, S, Sp, nh, p_drop=run.p_att, seed=sv.seed_att, klen=klen, border=border, row0=row0)
        sv.psave = sv.msave = None
    elif getattr(sv, "psave", None) is not None:
        # the training forward saved its probabilities: no recomputation of the scores (fbl_disent_attn_bwd_dsp)
        L.disent_attn_bwd_dsp(sv.psave, sv.msave, v, dctx, sv.lse, Dv, scale, dqkv[:, 2 * H:], dS, dST, B, S, Sp, nh,
                              p_drop=run.p_att, seed=sv.seed_att, klen=klen, border=border, row0=row0)
        sv.psave = sv.msave = None
    else:
        L.disent_attn_bwd_ds(q, k, v, dctx, pk, pq, relidx, run.mask_i32, sv.lse, Dv, scale, dqkv[:, 2 * H:], dS, dST,
                             B, S, Sp, nh, span2, p_drop=run.p_att, seed=sv.seed_att, klen=klen, border=border, lin=lin_a, row0=row0)
    m1 = m2 = None
    if use_gt:
        # G^T is k-blocked: [nh][B][Sp/32][rcnt][32] (every shear workgroup writes one contiguous block)
        if G1T is None:
            G1T = torch.empty(nh, B * (Sp // 32) * rcnt * 32, dtype=BF16, device=dev)
            G2T = torch.empty(nh, B * (Sp // 32) * rcnt * 32, dtype=BF16, device=dev)
        if POISON_GT:  # test switch: blocks the shear kernel legitimately leaves unwritten must never be read
            G1T.fill_(float("nan"))
            G2T.fill_(float("nan"))
        # (without klen the products read every row of G^T: everything outside the windows must then be zero-filled)
        m1, m2 = gt_tilemasks(eng, run) if klen is not None else (None, None)
    if toep_dq:
        L.disent_attn_bwd_dq(dS, k, PKX, dqkv[:, :H], B, S, Sp, nh, klen=klen, border=border, row0=row0)
    else:
        L.disent_attn_bwd_shear(0, dS, KT, PKT, relidx, dqkv[:, :H], G1T, B, S, Sp, nh, span2, klen=klen, rmin=rmin, rcnt=rcnt,
                                lin=lin, border=border, row0=row0, tilemask=m1)
    if not fused_dk:
        L.disent_attn_bwd_shear(1, dST, QT, PQT, relidx, dqkv[:, H:2 * H], G2T, B, S, Sp, nh, span2, klen=klen, rmin=rmin,
                                rcnt=rcnt, lin=lin, border=border, row0=row0, tilemask=m2)
    if use_gt:
        del dS, dST
        state = dict(G1T=G1T, G2T=G2T, QT=QT, KT=KT, rmin=rmin, rcnt=rcnt, B=B, Sp=Sp, klen=klen, masks=(m1, m2))
    else:
        # fbl_attn_pos_grad reads dS / dS^T and the token rows of q / k themselves
        state = dict(dS=dS, dST=dST, q=q, k=k, rmin=rmin, rcnt=rcnt, B=B, S=S, Sp=Sp, klen=klen, row0=row0)
    if defer_pos:
        return state
    dpos = pos_table_grads(eng, state, getattr(eng, "sk_ws", None))
    dpqk.copy_(dpos)  # fp32 -> bf16
    return None


def pos_table_grads(eng, st, ws):
    """dPK[h] = sum_b G1^T[h] . Q^T[h]^T,  dPQ[h] = sum_b G2^T[h] . K^T[h]^T  -> fp32 [span2, 2H] laid out [dPQ | dPK]
    (head h owns columns h*64 .. h*64+63); per-head split-K GEMMs on the k-blocked G^T."""
    H, nh, span2 = eng.H, eng.nh, eng.span2
    rmin, rcnt, B, Sp = st["rmin"], st["rcnt"], st["B"], st["Sp"]
    dpos = torch.zeros(span2, 2 * H, dtype=F32, device=eng.dev)
    if "dS" in st:  # one execution through the fused kernel: [nh, rcnt, 64] per table -> the [dPQ | dPK] column blocks
        dlo, dcnt, cmax = _delta_ranges(st["S"], eng.cfg, eng.dev)
        for neg, X, Y, col0 in ((0, st["dS"], st["q"], H), (1, st["dST"], st["k"], 0)):
            d = torch.empty(1, nh, rcnt, 64, dtype=F32, device=eng.dev)
            L.attn_pos_grad(neg, [X], [Y], dlo, dcnt, cmax, d, B, st["S"], Sp, nh, rcnt, klen=st["klen"], row0=st["row0"])
            dpos[rmin:rmin + rcnt, col0:col0 + H].view(rcnt, nh, 64).copy_(d[0].permute(1, 0, 2))
        return dpos
    G1T, G2T, QT, KT = st["G1T"], st["G2T"], st["QT"], st["KT"]
    Kc = B * Sp
    # split count: 5 at the bench shape (K = B*Sp = 10240).  Measured step time for 2 / 3 / 4 / 5 / 10 slices: 49.20 /
    # 48.90 / 48.89 / 48.56-48.71 / 48.88-48.95 ms (same box) -- half the partial-sum traffic of 10, still short workgroups
    sk = max(2, min(16, Kc // 2048), -(-(Kc // 64) // 2048))  # (the k-skipping path lists at most 2048 steps per slice)
    o_pk = torch.as_strided(dpos, (nh, rcnt, 64), (64, 2 * H, 1), H + rmin * 2 * H)
    o_pq = torch.as_strided(dpos, (nh, rcnt, 64), (64, 2 * H, 1), rmin * 2 * H)
    kblk = rcnt * 32  # elements between consecutive 32-wide k blocks of G^T
    a1 = torch.as_strided(G1T, (nh, rcnt, 32), (G1T.stride(0), 32, 1))
    a2 = torch.as_strided(G2T, (nh, rcnt, 32), (G2T.stride(0), 32, 1))
    # G^T blocks beyond a sample's last valid position are all zero: the shear kernel does not write them and the GEMM
    # skips those k-steps (roughly half of K on ragged batches)
    ks = dict(kskip_len=st["klen"], kskip_steps=Sp // 64) if st.get("klen") is not None else {}
    m1, m2 = st.get("masks", (None, None)) if ks else (None, None)
    L.gemm(a1, QT.view(nh, 64, Kc), out_f32=o_pk, splitk=sk, ws=ws, K=Kc, a_kblock=kblk, kskip_tilemask=m1, **ks)
    L.gemm(a2, KT.view(nh, 64, Kc), out_f32=o_pq, splitk=sk, ws=ws, K=Kc, a_kblock=kblk, kskip_tilemask=m2, **ks)
    return dpos


def pos_chain_buffers(eng, run, n_exec):
    """Per-step tensors that receive, execution by execution, what the position-table products of ALL layer executions need:
    G1^T / G2^T (k-blocked, written by the shear passes) and Q^T / K^T (written by the preparation kernel).  One allocation each,
    [n_exec, nh, ...]: execution e is slice e, and (execution, head) is ONE strided batch dimension for the GEMMs at the end."""
    B, S, nh = run.B, run.S, eng.nh
    Sp = (S + 63) // 64 * 64
    rmin, rcnt = _relidx_range(S, eng.cfg)
    blk = B * (Sp // 32) * rcnt * 32
    dev = eng.dev
    if not getattr(eng, "pos_grad_gt", False):
        # the fused kernel reads every execution's dS / dS^T (2 x 157 MB at the bench shape, kept until the end of backward:
        # the same 8 GB the G^T tensors took) and the token rows of its saved q / k
        return dict(X1=[], X2=[], Yq=[], Yk=[], rmin=rmin, rcnt=rcnt, B=B, S=S, Sp=Sp, n=0, seeds=[], cap=n_exec, klen=None, row0=None)
    return dict(G1T=torch.empty(n_exec, nh, blk, dtype=BF16, device=dev), G2T=torch.empty(n_exec, nh, blk, dtype=BF16, device=dev),
                QT=torch.empty(n_exec, nh, 64, B, Sp, dtype=BF16, device=dev), KT=torch.empty(n_exec, nh, 64, B, Sp, dtype=BF16, device=dev),
                rmin=rmin, rcnt=rcnt, B=B, Sp=Sp, n=0, seeds=[], cap=n_exec)


def pos_table_grads_batched(eng, run, pc):
    """The relative-position-table gradient of the whole backward pass in five launches at its END (it feeds only
    encoder.LayerNorm's gamma / beta, the last thing backward needs): until round 4 every layer execution ran its own chain
    (two split-K products, folds, cast, projection, dropout, accumulation) on a side stream next to the following layer's GEMMs
    -- 9.3 ms of side-queue kernel time per step whose long-lived workgroups cost the main stream 2.7 ms.  Now, over all E
    executions at once (a strided batch of E*nh problems each):
        dPK[e,h] = G1^T[e,h] . Q^T[e,h]^T ,  dPQ[e,h] = G2^T[e,h] . K^T[e,h]^T     (k-steps beyond a sample's length skipped)
        dR_e     = [dPQ | dPK]_e . [Wq ; Wk]_e                                       (one batched GEMM against the packed weights)
        dR       = sum_e dropout_e(dR_e)                                             (fbl_dropout_sum_f32: each through its own mask)
    Returns dR [span2, H] fp32.  autograd of model/deberta.py:779, 847-853, 870-918 summed over the executions."""
    H, nh, span2 = eng.H, eng.nh, eng.span2
    E, rmin, rcnt, B, Sp = pc["n"], pc["rmin"], pc["rcnt"], pc["B"], pc["Sp"]
    dev = eng.dev
    Kc = B * Sp
    kblk = rcnt * 32
    ks = dict(kskip_len=run.klen, kskip_steps=Sp // 64) if getattr(run, "klen", None) is not None else {}
    # [dPQ | dPK] of every execution, rows rmin .. rmin + rcnt of the tables (the others cannot be touched: their gradient is
    # zero): bf16 operand of the projection, fully written by the two copies below
    dpb = torch.empty(E, rcnt, 2 * H, dtype=BF16, device=dev)
    if "X1" in pc:
        dlo, dcnt, cmax = _delta_ranges(pc["S"], eng.cfg, dev)
        for neg, kx, ky, col0 in ((0, "X1", "Yq", H), (1, "X2", "Yk", 0)):
            d = torch.empty(E, nh, rcnt, 64, dtype=F32, device=dev)
            L.attn_pos_grad(neg, pc[kx][:E], pc[ky][:E], dlo, dcnt, cmax, d, B, pc["S"], Sp, nh, rcnt, klen=pc["klen"], row0=pc["row0"])
            L.heads_to_rows_bf16(d, dpb[:, :, col0:col0 + H])
        masks = ()
    else:
        masks = gt_tilemasks(eng, run) if ks else (None, None)
    for (key_g, key_t, col0), tmask in zip((("G1T", "QT", H), ("G2T", "KT", 0)), masks):
        G = pc[key_g][:E].view(E * nh, -1)
        a = torch.as_strided(G, (E * nh, rcnt, 32), (G.stride(0), 32, 1))
        T = pc[key_t][:E].view(E * nh, 64, Kc)
        d = L.zeros(E * nh, rcnt, 64, dtype=F32, device=dev)
        # two K slices (the skipping path is the accumulating one), folded deterministically through the workspace
        L.gemm(a, T, out_f32=d, splitk=max(2, -(-(Kc // 64) // 2048)), ws=eng.sk_ws, K=Kc, a_kblock=kblk, kskip_tilemask=tmask, **ks)
        # [e, h, r, 64] fp32 -> [e, r, h*64 + .] bf16, into this table's column block
        L.heads_to_rows_bf16(d.view(E, nh, rcnt, 64), dpb[:, :, col0:col0 + H])
    tmp = torch.empty(E, rcnt, H, dtype=F32, device=dev)
    L.gemm(dpb, eng.WposT_exec[:E], out_f32=tmp)
    dR = L.zeros(span2, H, dtype=F32, device=dev)
    # (dropout keys of the [span2, H] table: element (r, c) <-> r*H + c)
    L.dropout_sum_f32(tmp, pc["seeds"][:E] if run.p_hid > 0 else [0] * E, run.p_hid, dR[rmin:rmin + rcnt], key0=rmin * H)
    return dR
