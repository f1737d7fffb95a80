"""Plain-torch fp32 references of the individual ops (test infrastructure only; run on the GPU for speed)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def bf(x):
    """round-trip through bf16 (so kernel and reference see identical operand values)"""
    return x.to(torch.bfloat16).float()


def ref_attention(q, k, v, pk, pq, relidx, mask, scale, drop_keep=None):
    """q,k,v: [B,nh,S,64] fp32; pk,pq: [nh,R,64]; relidx int64 [2S-1]; mask [B,S] 0/1.
    Returns ctx [B,nh,S,64], lse [B,nh,S] (SURVEY App. C formula)."""
    B, nh, S, d = q.shape
    i = torch.arange(S, device=q.device)
    idx = relidx.long()[(i[:, None] - i[None, :]) + S - 1]  # [S,S]
    c2p = torch.einsum("bhid,hrd->bhir", q, pk)  # [B,nh,S,R]
    c2p = torch.gather(c2p, 3, idx[None, None].expand(B, nh, S, S))
    p2c = torch.einsum("bhjd,hrd->bhjr", k, pq)  # [B,nh,S(j),R]
    p2c = torch.gather(p2c, 3, idx.t()[None, None].expand(B, nh, S, S)).transpose(2, 3)  # [b,h,i,j] = KPQ[j, idx(i,j)]
    s = (torch.einsum("bhid,bhjd->bhij", q, k) + c2p + p2c) * scale
    m = mask.bool()
    m2 = m[:, None, :, None] & m[:, None, None, :]
    s = s.masked_fill(~m2, float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.softmax(s, -1)
    p = torch.where(m2, p, torch.zeros_like(p))
    p = torch.nan_to_num(p, nan=0.0)
    if drop_keep is not None:
        p = p * drop_keep
    return torch.einsum("bhij,bhjd->bhid", p, v), lse


def heads(x, B, S, nh):
    """[B*S, nh*64] -> [B,nh,S,64]"""
    return x.view(B, S, nh, 64).permute(0, 2, 1, 3)


def unheads(x):
    B, nh, S, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B * S, nh * d)


def stats(name, got, ref):
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    return (f"{name}: max_abs_err={err.max().item():.3e} mean_abs_err={err.mean().item():.3e} "
            f"ref_absmax={ref.abs().max().item():.3e} ref_std={ref.std().item():.3e}")
