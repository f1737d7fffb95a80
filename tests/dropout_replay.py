"""Rebuild, on the host, the dropout masks a HIP training forward drew (test infrastructure only).

The HIP path never stores a dropout mask: every seeded kernel derives its keep decisions from a counter-based hash of
(seed, element index) and the backward regenerates them (frozenbilm_amd/csrc/fbl_common.h fbl_hash / fbl_drop_thresh, attn_common.h "attention-
probability dropout RNG").  The hash is a pure function, so the masks of a finished forward can be recomputed here in numpy
uint64 / uint32 arithmetic from the per-site seeds the engine recorded (engine.Run / LayerSave) and handed to the CPU oracle
(oracle.dropout_masks) -- a train-mode comparison with the oracle then sees a dropout site applied at the wrong place, with the
wrong 1/(1-p), or missing in backward, which the keep-rate / self-consistency tests of the kernels cannot.

Key conventions restated (each with the kernel that owns it):
  * row sites -- embeddings (fbl_dropout_f32), block dropout in front of a LayerNorm (fbl_ln_fwd / the adapter tail
    fbl_adapter_up_resid_fwd), convolution branch (fbl_dropout_gelu_fwd), position table (fbl_dropout_f32 on [2*span, H]):
    element (m, n) of a [M, H] tensor is keyed by m*H + n;
  * adapter bottleneck (fbl_adapter_down_fwd / fbl_dense_adapter_down_fwd): element (m, a) keyed by m*ldz + a, ldz = row
    stride of the saved z;
  * attention probabilities (attn_fwd.hip / attn_bwd.hip kernel A): two 32-bit keys per (seed, sample*heads + head), one
    32-bit block hash per 2x2 block of (query, key) pairs, four 16-bit fields compared with a 16-bit threshold.
"""
from __future__ import annotations

import numpy as np
import torch

def fbl_hash(seed: int, idx: np.ndarray) -> np.ndarray:
    """fbl_common.h fbl_hash: index * odd constant, seed folded in by XOR, two multiply-xorshift rounds (uint32 wrap-around)"""
    idx = idx.astype(np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    s_lo, s_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        h = lo * np.uint32(0x9E3779B1)
        h ^= s_lo
        h ^= (hi << np.uint32(13)) | (hi >> np.uint32(19))
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h ^= s_hi
        h *= np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    return h


def drop_thresh(p: float) -> int:
    """fbl_common.h fbl_drop_thresh: p (a float) -> 32-bit threshold, drop iff hash < thresh"""
    t = float(np.float32(p)) * 4294967296.0
    return int(min(max(t, 0.0), 4294967295.0))


def row_mask(seed: int, shape, p: float, ld: int = None) -> torch.Tensor:
    """multiplicative mask (0 or 1/(1-p)) of a [.., cols] tensor whose element (m, n) is keyed by m*ld + n (ld = cols by default)"""
    cols = shape[-1]
    rows = int(np.prod(shape[:-1]))
    ld = cols if ld is None else ld
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(ld) + np.arange(cols, dtype=np.uint64)[None, :])
    keep = fbl_hash(seed, idx) >= np.uint32(drop_thresh(p))
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))  # the kernels' 1.0f / (1.0f - p)
    return torch.from_numpy(np.where(keep, inv, np.float32(0.0)).astype(np.float32)).view(*shape)


def attn_mask(seed: int, B: int, nh: int, S: int, p: float) -> torch.Tensor:
    """attn_common.h attn_drop_key / attn_drop_block / attn_drop_keep -> [B, nh, S, S] multiplicative mask"""
    Sp = (S + 63) // 64 * 64
    Sp2 = Sp // 2
    t = np.float32(p) * np.float32(65536.0) + np.float32(0.5)
    thr16 = np.uint32(min(float(t), 65535.0)) if p > 0 else np.uint32(0)
    inv = np.float32(65536.0) / (np.float32(65536.0) - np.float32(thr16))
    bh = np.arange(B * nh, dtype=np.uint64)
    k1 = fbl_hash(seed, 2 * bh)[:, None, None]
    k2 = fbl_hash(seed, 2 * bh + np.uint64(1))[:, None, None]
    i = np.arange(S, dtype=np.uint32)
    blk = ((i[:, None] >> 1) * np.uint32(Sp2) + (i[None, :] >> 1)).astype(np.uint32)[None]  # [1, S, S]
    with np.errstate(over="ignore"):
        x = blk * np.uint32(0x9E3779B1) + k1
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
        y = (x ^ k2) * np.uint32(0x2C1B3C6D)
        y ^= y >> np.uint32(15)
    pi = (i & 1).astype(bool)[None, :, None]
    pj = (i & 1).astype(bool)[None, None, :]
    w = np.where(pi, y, x)
    f = np.where(pj, w >> np.uint32(16), w & np.uint32(0xFFFF))
    keep = f >= thr16
    return torch.from_numpy(np.where(keep, inv, np.float32(0.0)).astype(np.float32)).view(B, nh, S, S)


class ReplayedMasks:
    """Provider for oracle.dropout_masks built from a finished HIP training forward (`out._run`, BEFORE backward pops the
    saved layer executions).  The oracle asks in the reference's execution order: "emb", then per layer execution "pos", "att",
    "ad", "hid", "ad", "hid" (adapters only where the model has them), "conv" after layer 0.  `seed_word`: value of the device
    seed word the launches added to their per-site seeds (captured steps; 0 for eager launches)."""

    def __init__(self, run, cfg, nh: int, p_hid: float, p_att: float, p_ad: float, seed_word: int = 0):
        self.B, self.S, self.nh = run.B, run.S, nh
        self.p_hid, self.p_att, self.p_ad = p_hid, p_att, p_ad
        w = seed_word

        def s(v):
            return (int(v) + w) & 0xFFFFFFFFFFFFFFFF

        self.seed_emb, self.seed_conv = s(run.seed_emb), s(getattr(run, "seed_conv", 0))
        self.execs = [dict(pos=s(sv.seed_pos), att=s(sv.seed_att), ad=[s(sv.seed_ad1), s(sv.seed_ad2)],
                           hid=[s(sv.seed_ln1), s(sv.seed_ln2)],
                           ldz=[sv.z1.stride(0) if sv.z1 is not None else 0, sv.z2.stride(0) if sv.z2 is not None else 0])
                      for sv in run.layers]
        self.has_ad = [cfg.ds_factor_attn > 0, cfg.ds_factor_ff > 0]
        self.e_pos = self.e_att = 0
        self.e_site = 0  # 2 * exec + site for "hid"
        self.e_ad = 0
        self.asked = []

    def __call__(self, kind, shape):
        self.asked.append(kind)
        if kind == "emb":
            return row_mask(self.seed_emb, shape, self.p_hid)
        if kind == "conv":
            return row_mask(self.seed_conv, shape, self.p_hid)
        if kind == "pos":
            e = self.execs[self.e_pos]
            self.e_pos += 1
            return row_mask(e["pos"], shape, self.p_hid)
        if kind == "att":
            e = self.execs[self.e_att]
            self.e_att += 1
            B, nh, S, S2 = shape
            assert (B, nh, S, S2) == (self.B, self.nh, self.S, self.S)
            return attn_mask(e["att"], B, nh, S, self.p_att)
        if kind == "hid":
            e, site = self.execs[self.e_site // 2], self.e_site % 2
            self.e_site += 1
            return row_mask(e["hid"][site], shape, self.p_hid)
        if kind == "ad":
            sites = [i for i in (0, 1) if self.has_ad[i]]
            e, site = self.execs[self.e_ad // len(sites)], sites[self.e_ad % len(sites)]
            self.e_ad += 1
            return row_mask(e["ad"][site], shape, self.p_ad, ld=e["ldz"][site])
        raise KeyError(kind)

    def exhausted(self) -> bool:
        n = len(self.execs)
        return self.e_pos == n and self.e_att == n and self.e_site == 2 * n and self.e_ad == n * sum(self.has_ad)
