"""Explicit forward / backward pipelines of the FrozenBiLM masked-LM hot path on MI355X.

No tracing compiler and no ATen math: the step is a fixed sequence of C-ABI kernel launches (lib.py -> libfbl.so)
on the current HIP stream; torch only owns HBM allocations.  Semantics follow SURVEY.md App. C / the reference lines
quoted per stage.  Precision: bf16 MFMA operands, fp32 accumulation, fp32 residual stream / LayerNorm statistics /
softmax / logits / gradients of the trainable parameters.

Residual stream representation: a LayerNorm output is kept as (t, stats, gamma, beta[, rowmask]) -- the pre-norm
fp32 tensor plus per-row (mean, rstd) -- together with its bf16 copy (the next GEMM operand).  Consumers re-normalise
on the fly, so the fp32 normalised tensor is never written to HBM, and ``t`` doubles as the tensor saved for backward.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import lib as L
from .model.relpos import rel_index_vector

BF16, F32 = torch.bfloat16, torch.float32


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class NormRef:
    """A tensor given in LayerNorm-normalised form (see module docstring)."""
    t: torch.Tensor
    stats: torch.Tensor
    gamma: torch.Tensor
    beta: torch.Tensor
    rowmask: Optional[torch.Tensor] = None

    def as_args(self):
        return (self.t, self.stats, self.gamma, self.beta, self.rowmask)


@dataclass
class Stream:
    """One activation stream: bf16 GEMM operand + its fp32 value as NormRef or plain tensor."""
    bf16: torch.Tensor
    norm: Optional[NormRef] = None
    plain: Optional[torch.Tensor] = None
    full: Optional[torch.Tensor] = None  # [N + 2*span, H] buffer whose first N rows are `bf16`: the tail receives the
    #                                      relative-position table R so that one GEMM yields Q|K|V and PQ|PK


@dataclass
class Packing:
    """Row layout of a ragged batch without its trailing padding (opt-in `model.packed_rows`): sample b owns the activation
    rows [row0[b], row0[b+1]) = its positions 0 .. plen[b]-1, where plen[b] - 1 is the last position that anything reads
    (valid token, label, requested logit row; at least the video slots).  Every kernel between the embedding and the
    head is row-wise except the attention kernels (which take row0, include/fbl.h) and three once-per-step stages that
    go through the padded grid (embedding gather, the k=3 convolution's im2col, the position embeddings of the decoder)."""
    row0: torch.Tensor   # int32 [B+1]
    sel: torch.Tensor    # int64 [Np]: padded row b*S+s of each packed row
    inv: torch.Tensor    # int64 [B*S]: packed row of each padded row, -1 where it has none
    pos: torch.Tensor    # int64 [Np]: position s of each packed row
    n: int               # Np


class Engine:
    def __init__(self, model):
        self.m = model
        cfg = model.config
        self.cfg = cfg
        self.H, self.I, self.V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        self.nh = cfg.num_attention_heads
        self.nL = cfg.num_hidden_layers
        self.span2 = 2 * cfg.att_span
        # |i - j| < lin_span: identity buckets of the relative-position map (model/deberta.py:578-589: mid = bucket_size // 2)
        self.lin_span = min(cfg.position_buckets // 2, cfg.att_span) if cfg.position_buckets > 0 else 0
        self.F = model.features_dim
        self.Fp = _ru(self.F, 64) if self.F else 0
        self.A1 = self.H // model.ds_factor_attn if model.ds_factor_attn else 0
        self.A2 = self.H // model.ds_factor_ff if model.ds_factor_ff else 0
        self.Vp = _ru(self.V, 64)
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("frozenbilm_amd runs on MI355X only: move the model to a cuda (HIP) device; "
                               "there is no CPU fallback")
        L.load()
        self.dev = dev
        self.opts = dict(getattr(model, "engine_options", None) or {})
        unknown = set(self.opts) - {"eager_logits", "side_stream", "fold_dx", "dw_on_side", "fuse_tail", "merge", "dw_group", "attn_save_p",
                                    "pos_grad_gt", "attn_fused_dk", "attn_toeplitz_dq"}
        if unknown:
            raise ValueError(f"unknown engine_options: {sorted(unknown)}")
        self._build_flat()
        self._pack_frozen()
        self._relidx: Dict[int, torch.Tensor] = {}
        self._pos_cache: Dict[int, torch.Tensor] = {}  # per-layer position projections of inference forwards (_pos_proj_cached)
        self._no_pos_cache = False  # set while an inference graph is captured: a replay must not depend on state kept outside it
        self.reducer = None  # set by parallel.GradReducer for data-parallel training
        self._pad_bufs = {}
        self._dyz_pool: Dict[tuple, list] = {}  # [N, K_fold] operands of the folded adapter backward with their padding zeroed once
        self._dyz_shape = None                   # ... of the current batch shape only
        self.params_version = 0  # bumped by FusedAdam.step (which updates the flat buffer through raw pointers)
        self._ops_version = None
        self.skip_dead_layer = True
        self._ln_ws = L.ln_bwd_ws(self.H, dev)
        self._cs_ws = L.colsum_ws(max(self.H, self.I), dev)
        # split-K partials (main stream): 192 MiB, enough for the prediction-head backward of ~2000 labelled rows (15 slices x
        # rows x H floats) -- with a smaller workspace that GEMM falls back to atomic adds and the step stops being
        # reproducible bit for bit
        self.sk_ws = torch.empty(48 << 20, dtype=F32, device=dev)
        # trainable-weight gradients are off the critical path (nothing downstream in backward reads them): they run on a
        # side HIP stream and fill the tails of the big dX GEMMs; own workspaces so they never race with the main stream
        self.side = torch.cuda.Stream(device=dev)  # (stream priorities were measured: no effect on the interference)
        # the caller-provided aux stream of the GEMM entry points (remainder rows of multi-round problems run there,
        # concurrently with the big tiles; include/fbl.h): one per device, owned by the host side
        if L._AUX.get(dev.index if dev.index is not None else torch.cuda.current_device()) is None:
            L.set_aux_stream(torch.cuda.Stream(device=dev), dev)
        # Engine options: `model.engine_options` (a dict read ONCE, when the engine is built; the product reads no environment
        # variable).  Defaults are the shipped configuration; the other values exist for A/B measurements (tools/, tests):
        #   eager_logits   fill the full [N, V] logits in every forward even when only the loss is consumed (reference-eager)
        #   side_stream    False = single-stream execution (the per-step composition of the merged adapter rows otherwise runs beside the forward)
        #   fold_dx        adapter dx folded into the dense dX GEMM;  dw_on_side: generic dW route on the side stream
        #   fuse_tail      adapter up-projection + block dropout + residual as one epilogue (fbl_adapter_up_resid_fwd)
        #   merge          dense layer + adapter down-projection as one GEMM;  dw_group: adapter gradient products per launch
        o = self.opts
        self.eager_logits = bool(o.get("eager_logits", False))
        self.side_ws = torch.empty(8 << 20, dtype=F32, device=dev)
        self.side_cs_ws = L.colsum_ws(max(self.H, self.I), dev)
        self.use_side_stream = bool(o.get("side_stream", True))
        self.fold_dx = bool(o.get("fold_dx", True))
        self.dw_on_side = bool(o.get("dw_on_side", False))
        #   attn_save_p    False = the attention backward recomputes the probabilities (rounds 1-5) instead of reading the ones the
        #                  training forward saved (157 MB per layer execution at the bench shape, held until the backward)
        self.attn_save_p = bool(o.get("attn_save_p", True))
        #   pos_grad_gt    True = position-table gradients through G^T and split-K GEMMs (rounds 1-5) instead of fbl_attn_pos_grad
        self.pos_grad_gt = bool(o.get("pos_grad_gt", False))
        #   attn_fused_dk  False = dK by the separate key-major shear pass (rounds 1-5) instead of inside kernel A (fbl_disent_attn_bwd_dspk)
        self.attn_fused_dk = bool(o.get("attn_fused_dk", True))
        #   attn_toeplitz_dq  False = dQ by the scatter-based query-major shear pass (rounds 1-5) instead of fbl_disent_attn_bwd_dq
        self.attn_toeplitz_dq = bool(o.get("attn_toeplitz_dq", True))
        self.fuse_tail = bool(o.get("fuse_tail", True))
        L.exclude_from_aux(self.side)  # side-stream GEMMs never fork into the aux stream of the main stream's GEMMs
        self.dw_group = max(1, min(L.ADW_MAX_ADAPTERS, int(o.get("dw_group", 16))))  # adapter gradient products per launch (<= 16)

    # ------------------------------------------------------------------ parameter plumbing
    def _build_flat(self):
        """Re-home every trainable parameter into one flat fp32 buffer (and its grad into a flat grad buffer)."""
        from .model.deberta import flat_order

        m = self.m
        named = dict(m.named_parameters())
        train_names = [n for n, p in named.items() if p.requires_grad]
        order = flat_order(self.cfg, train_names)
        offs, total = {}, 0
        for n in order:
            offs[n] = total
            total += _ru(named[n].numel(), 8)  # keep every view 32-byte aligned
        self.flat = torch.zeros(total, dtype=F32, device=self.dev)
        # the gradient buffer is preceded by a few floats that travel with the first data-parallel bucket (the step's logged
        # loss: parallel.GradReducer.stage_scalars); clip + Adam and p.grad only ever see `flat_grad`
        from .parallel import SCALAR_SLOT

        self.flat_grad_full = torch.zeros(SCALAR_SLOT + total, dtype=F32, device=self.dev)
        self.flat_grad = self.flat_grad_full[SCALAR_SLOT:]
        self.flat_bf16 = torch.zeros(total, dtype=BF16, device=self.dev)
        self.offsets, self.order = offs, order
        self._adT_offs = {}
        self.G: Dict[str, torch.Tensor] = {}
        self.Pb: Dict[str, torch.Tensor] = {}
        for n in order:
            p = named[n]
            o, k = offs[n], p.numel()
            view = self.flat[o:o + k].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.G[n] = self.flat_grad[o:o + k].view(p.shape)
            self.Pb[n] = self.flat_bf16[o:o + k].view(p.shape)
        self.P = {n: p.data for n, p in named.items()}
        self.named = named
        # bucket boundaries (one bucket per backward stage) for the gradient all-reduce
        self.bucket_ends: Dict[str, int] = {}
        for n in order:
            key = self._bucket_key(n)
            self.bucket_ends[key] = offs[n] + _ru(named[n].numel(), 8)

    def _bucket_key(self, name: str) -> str:
        if name.startswith("lm_predictions."):
            return "head"
        if name.startswith("deberta.encoder.layer."):
            return "layer" + name.split(".")[3]
        if name.startswith("deberta.encoder.conv."):
            return "layer0" if False else "conv"
        if name.startswith("deberta.encoder.LayerNorm"):
            return "relln"
        return "emb"

    def attach_grads(self):
        """Make p.grad the views of the flat grad buffer; zero it when the user dropped the grads (set_to_none)."""
        fresh = all(self.named[n].grad is None for n in self.order)
        if fresh:
            L.zero_(self.flat_grad)
        for n in self.order:
            p = self.named[n]
            if p.grad is None:
                if not fresh:
                    self.G[n].zero_()
                p.grad = self.G[n]
            elif p.grad.data_ptr() != self.G[n].data_ptr():
                self.G[n].copy_(p.grad)
                p.grad = self.G[n]

    def _pack_frozen(self):
        P, H, I = self.P, self.H, self.I
        nL, dev = self.nL, self.dev
        bf = lambda t: t.to(BF16).contiguous()
        # A dense layer followed by an adapter runs as ONE GEMM against [W ; Wd.W] (fbl_dense_adapter_down_fwd): the
        # composed rows Wd.W change with every optimizer step and are rebuilt by two batched GEMMs per site type
        # (refresh_trainable_operands).  For those, the transposed frozen weights of all layers live in one tensor, in
        # REVERSE layer order (index j = nL-1-layer) -- the order of the adapters in the flat trainable buffer.
        # (the merged GEMM needs the bottleneck unpadded, A % 64 == 0, and the segment boundary H on a wave's column
        # range, H % 64 == 0: include/fbl.h; otherwise neither the composed weights nor their per-step rebuild exist)
        self.merge1 = bool(self.A1) and self.A1 % 64 == 0 and H % 64 == 0 and bool(self.opts.get("merge", True))
        self.merge2 = bool(self.A2) and self.A2 % 64 == 0 and H % 64 == 0 and bool(self.opts.get("merge", True))
        self.WoT_rev = torch.empty(nL, H, H, dtype=BF16, device=dev)
        self.WdT_rev = torch.empty(nL, I, H, dtype=BF16, device=dev)
        if self.merge1:
            # backward twin of the merged weights: dctx = [dy | dz] . [Wo^T | (Wd.Wo)^T]^T folds the adapter's dx = dy + dz.Wd
            # into the dense layer's dX GEMM (K = H + A1, padded to an even number of 64-wide K tiles with zero columns)
            self.Kf1 = _ru(H + self.A1, 128)
            self.WoF_rev = torch.zeros(nL, H, self.Kf1, dtype=BF16, device=dev)
            self.WoM_rev = torch.zeros(nL, H + self.A1, H, dtype=BF16, device=dev)
            self.boM_rev = torch.zeros(nL, H + self.A1, dtype=F32, device=dev)
            self.bo16_rev = torch.zeros(nL, 1, H, dtype=BF16, device=dev)
        if self.merge2:
            self.WdM_rev = torch.zeros(nL, H + self.A2, I, dtype=BF16, device=dev)
            self.bdM_rev = torch.zeros(nL, H + self.A2, dtype=F32, device=dev)
            self.bd16_rev = torch.zeros(nL, 1, H, dtype=BF16, device=dev)
        self.Lw = []
        for i in range(self.nL):
            p = f"deberta.encoder.layer.{i}"
            s = p + ".attention.self."
            j = nL - 1 - i
            Wqkv = torch.cat([P[s + "query_proj.weight"], P[s + "key_proj.weight"], P[s + "value_proj.weight"]], 0)
            self.WoT_rev[j].copy_(P[p + ".attention.output.dense.weight"].t())
            self.WdT_rev[j].copy_(P[p + ".output.dense.weight"].t())
            d = dict(
                Wqkv=bf(Wqkv), WqkvT=bf(Wqkv.t()),
                bqkv=torch.cat([P[s + "query_proj.bias"], P[s + "key_proj.bias"], P[s + "value_proj.bias"]]).float().contiguous(),
                Wo=bf(P[p + ".attention.output.dense.weight"]), WoT=self.WoT_rev[j],
                bo=P[p + ".attention.output.dense.bias"].float().contiguous(),
                Wi=bf(P[p + ".intermediate.dense.weight"]), WiT=bf(P[p + ".intermediate.dense.weight"].t()),
                bi=P[p + ".intermediate.dense.bias"].float().contiguous(),
                Wd=bf(P[p + ".output.dense.weight"]), WdT=self.WdT_rev[j],
                bd=P[p + ".output.dense.bias"].float().contiguous(),
            )
            if self.merge1:
                self.WoM_rev[j, :H].copy_(d["Wo"])
                self.boM_rev[j, :H].copy_(d["bo"])
                self.bo16_rev[j, 0].copy_(d["bo"])
                d["WoM"], d["boM"] = self.WoM_rev[j], self.boM_rev[j]
                self.WoF_rev[j, :, :H].copy_(self.WoT_rev[j])
                d["WoF"] = self.WoF_rev[j]
            if self.merge2:
                self.WdM_rev[j, :H].copy_(d["Wd"])
                self.bdM_rev[j, :H].copy_(d["bd"])
                self.bd16_rev[j, 0].copy_(d["bd"])
                d["WdM"], d["bdM"] = self.WdM_rev[j], self.bdM_rev[j]
            self.Lw.append(d)
        # [Wq ; Wk]^T of every layer EXECUTION that has a backward, in backward order (the last layer's two enhanced-mask-decoder
        # passes, then layers nL-2 .. 0): the position-table gradients of all of them are projected by one strided-batch GEMM
        order = [nL - 1, nL - 1] + list(range(nL - 2, -1, -1))
        self.WposT_exec = torch.empty(len(order), H, 2 * H, dtype=BF16, device=dev)
        for e, li in enumerate(order):
            self.WposT_exec[e].copy_(self.Lw[li]["WqkvT"][:, : 2 * H])
        if self.cfg.conv_kernel_size:
            w = P["deberta.encoder.conv.conv.weight"]  # [H_out, H_in, 3] -> [H_out, k*H_in + c]
            W2 = w.permute(0, 2, 1).reshape(H, 3 * H)
            self.Wc, self.WcT = bf(W2), bf(W2.t())
            self.bc = P["deberta.encoder.conv.conv.bias"].float().contiguous()
        hd = "lm_predictions.lm_head."
        self.Wh, self.WhT = bf(P[hd + "dense.weight"]), bf(P[hd + "dense.weight"].t())
        self.bh = P[hd + "dense.bias"].float().contiguous()
        E = P["deberta.embeddings.word_embeddings.weight"]
        self.E32 = E.float().contiguous()
        self.Eb = bf(E)
        self.ETb = torch.zeros(H, self.Vp, dtype=BF16, device=self.dev)
        self.ETb[:, : self.V] = E.t().to(BF16)
        self.head_bias = P[hd + "bias"].float().contiguous()
        self.rel_emb = P["deberta.encoder.rel_embeddings.weight"].float().contiguous()
        self.pos_emb = P["deberta.embeddings.position_embeddings.weight"].float().contiguous()
        if self.m.n_ans:
            T = P["answer_embeddings.weight"]
            self.n_ans = T.shape[0]
            self.Ansb = bf(T)
            self.ans_bias = P["answer_bias"].float().contiguous()

    def relidx(self, S: int) -> torch.Tensor:
        if S not in self._relidx:
            v = rel_index_vector(S, self.cfg.position_buckets, self.cfg.max_rel, self.cfg.att_span)
            self._relidx[S] = torch.from_numpy(np.ascontiguousarray(v)).to(self.dev)
        return self._relidx[S]

    def _refresh_if_stale(self, need_grad: bool):
        """bf16 operands / composed adapter rows follow the trainable parameters: rebuilt before EVERY forward by default
        (0.2 ms of GPU time).  Only inside `model.weights_frozen()` -- the evaluate loops, the inference-graph path -- is the
        rebuild skipped while nothing the engine can see has written to them: FusedAdam.step (params_version) and in-place
        updates through the parameter objects (autograd version counters).  Writes through `.data`, into `engine.flat` or by
        a broadcast into the flat buffer bump neither counter, which is why skipping is opt-in."""
        frozen = getattr(self.m, "_weights_frozen", 0) > 0
        ver = (self.params_version, tuple(self.named[n]._version for n in self.order)) if (frozen and not need_grad) else None
        if ver is None or ver != self._ops_version:
            self.refresh_trainable_operands()
            self._ops_version = ver

    def seed_word_value(self) -> int:
        """the 64-bit word a training pass adds to its per-site dropout seeds: the model's position in its mask stream"""
        return (self.m.dropout_seed_base() * 0x9E3779B1) & 0x7FFFFFFFFFFFFFFF

    def prepare_inference(self):
        """Everything an inference forward needs from OUTSIDE its launch sequence, done now: the bf16 operand copies /
        composed adapter rows are current (rebuilt in place: a captured graph keeps reading the same buffers) and the
        current stream has waited for the side stream that composes them (a captured forward may not wait on events
        recorded outside its capture, so `_compose_ev` is consumed here)."""
        self._refresh_if_stale(False)
        ev = getattr(self, "_compose_ev", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev[1])
            self._compose_ev = None

    def invalidate_operands(self):
        """the trainable parameters were modified in a way the engine cannot see (through `.data`): rebuild on next use"""
        self._ops_version = None
        self.params_version += 1

    def refresh_trainable_operands(self):
        """bf16 MFMA operands of the trainable matrices (they change every optimizer step)."""
        self._pos_cache.clear()
        L.cast_bf16(self.flat, self.flat_bf16)
        H = self.H
        self.ad = []
        for i in range(self.nL):
            p = f"deberta.encoder.layer.{i}"
            ent = {}
            for key, blk, A in (("a1", ".attention.output.adapter", self.A1), ("a2", ".output.adapter", self.A2)):
                if not A:
                    continue
                Ap = _ru(A, 64)
                down = self.Pb[p + blk + ".down.weight"]  # [A,H]
                up = self.Pb[p + blk + ".up.weight"]  # [H,A]
                if Ap != A:  # zero-padded copy in a persistent buffer (stable address: captured graphs keep reading it)
                    upp = self._pad_bufs.get((i, key))
                    if upp is None:
                        upp = self._pad_bufs[(i, key)] = torch.zeros(H, Ap, dtype=BF16, device=self.dev)
                    upp[:, :A] = up
                    up = upp
                ent[key] = dict(down=down, up=up, A=A, Ap=Ap, name=p + blk,
                                bd=self.P[p + blk + ".down.bias"], bu=self.P[p + blk + ".up.bias"])
            self.ad.append(ent)
        # composed rows of the merged dense + down-projection weights: off the critical path, on the side stream (layer 0
        # first -- the forward reaches its merged GEMM after ~0.3 ms -- then all other layers in one strided batch)
        if self.merge1 or self.merge2:
            if self.use_side_stream:
                self.side.wait_stream(torch.cuda.current_stream())  # the bf16 cast of the flat buffer above
                with torch.cuda.stream(self.side):
                    self._compose_ev = self._compose_adapter_down(events=True)
            else:
                self._compose_adapter_down()
                self._compose_ev = None
        if self.F:
            wv = self.Pb["deberta.embeddings.linear_video.weight"]
            if self.Fp != self.F:
                w2 = self._pad_bufs.get("Wv")
                if w2 is None:
                    w2 = self._pad_bufs["Wv"] = torch.zeros(H, self.Fp, dtype=BF16, device=self.dev)
                w2[:, : self.F] = wv
                wv = w2
            self.Wv = wv

    def _compose_adapter_down(self, events=False, grad=True):
        """Rows [H, H+A) of the merged weights / biases: Wd.W and Wd.b + bd for every layer (see _pack_frozen), from the
        current adapter weights.  Layers nL-1 .. 1 sit at a constant stride in the flat trainable buffer: one strided-batch
        GEMM each for the matrix and the bias; layer 0 (the conv LayerNorm sits between it and layer 1) gets its own and
        goes first.  events=True: returns (event after layer 0, event after everything) recorded on the current stream."""
        H, nL = self.H, self.nL
        sites = []
        for on, blk, A, WT, WM, bM, b16, WF in ((self.merge1, ".attention.output.adapter", self.A1, self.WoT_rev,
                                                 getattr(self, "WoM_rev", None), getattr(self, "boM_rev", None),
                                                 getattr(self, "bo16_rev", None), getattr(self, "WoF_rev", None)),
                                                (self.merge2, ".output.adapter", self.A2, self.WdT_rev,
                                                 getattr(self, "WdM_rev", None), getattr(self, "bdM_rev", None),
                                                 getattr(self, "bd16_rev", None), None)):
            if not on:
                continue
            ow = [self.offsets[f"deberta.encoder.layer.{i}{blk}.down.weight"] for i in range(nL)]
            ob = [self.offsets[f"deberta.encoder.layer.{i}{blk}.down.bias"] for i in range(nL)]
            if nL >= 3 and len({ow[i - 1] - ow[i] for i in range(2, nL)}) == 1 and \
                    len({ob[i - 1] - ob[i] for i in range(2, nL)}) == 1 and ow[nL - 2] - ow[nL - 1] == ob[nL - 2] - ob[nL - 1] > 0:
                groups = [(nL - 1, 1, ow[0], ob[0], A * H), (0, nL - 1, ow[nL - 1], ob[nL - 1], ow[nL - 2] - ow[nL - 1])]
            else:
                groups = [(nL - 1 - i, 1, ow[i], ob[i], A * H) for i in range(nL)]
            sites.append((A, WT, WM, bM, b16, groups, WF))

        def run_group(A, WT, WM, bM, b16, grp, WF=None):
            j0, nb, o_w, o_b, st = grp
            wd3 = torch.as_strided(self.flat_bf16, (nb, A, H), (st, H, 1), o_w)
            L.gemm(wd3, WT[j0:j0 + nb], out_bf16=WM[j0:j0 + nb, H:H + A, :])                # Wd . W
            if WF is not None and grad:  # (Wd . W)^T next to W^T: the K-extension of the folded backward GEMM
                L.gemm(WT[j0:j0 + nb], wd3, out_bf16=WF[j0:j0 + nb, :, H:H + A])
            bd3 = torch.as_strided(self.flat, (nb, A, 1), (st, 1, 1), o_b)
            bo3 = torch.as_strided(bM, (nb, A, 1), (bM.stride(0), 1, 1), bM[j0, H:].storage_offset())
            L.gemm(wd3, b16[j0:j0 + nb], aux=bd3, aux_kind=L.AUX_ADD_F32, out_f32=bo3)      # Wd . b + bd

        for A, WT, WM, bM, b16, groups, WF in sites:  # layer 0 of every site first
            run_group(A, WT, WM, bM, b16, groups[0], WF)
        ev0 = ev1 = None
        if events:
            ev0 = torch.cuda.Event()
            ev0.record()
        for A, WT, WM, bM, b16, groups, WF in sites:
            for grp in groups[1:]:
                run_group(A, WT, WM, bM, b16, grp, WF)
        if events:
            ev1 = torch.cuda.Event()
            ev1.record()
            return ev0, ev1
        return None

    def _adapter_bwd_operands(self, ent):
        """W^T operands for the adapter backward (trainable, so rebuilt after every optimizer step): all adapters of a
        shape are transposed by ONE batched launch straight out of the flat bf16 parameter copy, on first use in a
        step.  Bottlenecks that are not a multiple of 64 take the zero-padded per-adapter path."""
        if "upT" not in ent:
            H = self.H
            groups = {}
            for e in self.ad:
                for key in ("a1", "a2"):
                    if key in e:
                        groups.setdefault((e[key]["A"], e[key]["Ap"]), []).append(e[key])
            for (A, Ap), ents in groups.items():
                upT_all = torch.empty(len(ents), A, H, dtype=BF16, device=self.dev)
                for i, e in enumerate(ents):
                    e["upT"] = upT_all[i]  # [A,H] = up.weight[H,A]^T
                cache = self._adT_offs.get((A, Ap))
                if cache is None:
                    i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=self.dev)
                    cache = dict(up_src=i64([self.offsets[e["name"] + ".up.weight"] for e in ents]),
                                 dn_src=i64([self.offsets[e["name"] + ".down.weight"] for e in ents]),
                                 dst=i64([i * A * H for i in range(len(ents))]))
                    self._adT_offs[(A, Ap)] = cache
                L.transpose_batched_bf16(self.flat_bf16, cache["up_src"], upT_all, cache["dst"], H, A)
                if Ap == A:
                    dnT_all = torch.empty(len(ents), H, A, dtype=BF16, device=self.dev)
                    L.transpose_batched_bf16(self.flat_bf16, cache["dn_src"], dnT_all, cache["dst"], A, H)
                    for i, e in enumerate(ents):
                        e["downT"] = dnT_all[i]  # [H,A] = down.weight[A,H]^T
                else:  # K of the consuming GEMM must be a multiple of 64: zero-padded columns
                    for e in ents:
                        downT = torch.zeros(H, Ap, dtype=BF16, device=self.dev)
                        downT[:, :A].copy_(self.Pb[e["name"] + ".down.weight"].t())
                        e["downT"] = downT
        return ent["upT"], ent["downT"]

    # ------------------------------------------------------------------ public entry
    def run(self, input_ids, attention_mask, video, video_mask, labels, mlm, want_hidden, logit_rows=None, want_attn=False):
        m = self.m
        train = m.training
        need_grad = torch.is_grad_enabled() and any(self.named[n].requires_grad for n in self.order)
        if input_ids.device != self.dev:
            raise RuntimeError(f"inputs must be on {self.dev}")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        B, Lt = input_ids.shape
        use_video = bool(self.F) and video is not None
        T = video.shape[1] if use_video else 0
        S = T + Lt
        if S > self.cfg.max_position_embeddings:
            raise RuntimeError(
                f"sequence of {S} positions (video {T} + text {Lt}) exceeds max_position_embeddings="
                f"{self.cfg.max_position_embeddings} (the reference fails the same way, model/deberta.py:1020-1021,1392)")
        if use_video:
            if video_mask is None:
                video_mask = torch.ones(video.shape[:2], device=self.dev, dtype=attention_mask.dtype)
            mask = torch.cat([video_mask.to(attention_mask.dtype), attention_mask], 1)
        else:
            mask = attention_mask
        mask = mask.to(torch.int32).contiguous()
        full_labels = None
        if labels is not None:
            if use_video:
                full_labels = torch.cat([torch.full((B, T), -100, dtype=torch.long, device=self.dev), labels], 1)
            else:
                full_labels = labels
            full_labels = full_labels.contiguous().view(-1)
            # The labelled rows are a function of the INPUT only: find them before anything is queued, so the host
            # synchronisation inside nonzero() waits for the previous step at most, never for this forward.
            rows_labelled = torch.nonzero(full_labels != -100).view(-1)
        # Packed rows (opt-in, frozenbilm_amd extension): drop the trailing padding rows of every sample.  Only for calls
        # whose outputs live on selected rows (a loss, logit_rows); like the label rows, the layout is a function of the
        # INPUT, read back before anything is queued.
        pk = None
        if (getattr(m, "packed_rows", False) and not want_attn and (full_labels is not None or logit_rows is not None)
                and not torch.cuda.is_current_stream_capturing()):
            pk = self._make_packing(mask, full_labels, logit_rows, B, S, T)
        if train:
            m.step_seed += 1
        # Dropout seeds = a per-site constant (a function of the site index only) + the position of this step in the model's
        # mask stream.  Eager launches add the two on the host (Run.next_seed); a captured launch sequence keeps the per-site
        # constants as kernel arguments and reads the position from a device word (train_graph.py; include/fbl.h "Dropout
        # seeds") -- the same sum, so eager and replayed steps draw identical masks.
        run = Run(B=B, S=S, T=T, Lt=Lt, train=train, save=need_grad, seed_base=self.seed_word_value() if train else 0,
                  p_hid=self.cfg.hidden_dropout_prob if train else 0.0,
                  p_att=self.cfg.attention_probs_dropout_prob if train else 0.0,
                  p_ad=m.adapter_dropout if train else 0.0)
        run.mask = mask.view(-1)
        run.pk = pk
        run.N = pk.n if pk is not None else B * S
        run.want_attn = bool(want_attn)
        if want_attn and train and run.p_att > 0:
            raise NotImplementedError("output_attentions=True is served in eval mode (the probabilities of a training forward "
                                      "carry the attention dropout mask, which the fused kernel regenerates instead of storing)")
        run.labels = full_labels
        run.rows = rows_labelled if full_labels is not None else None
        if pk is not None and full_labels is not None:  # the head works on packed rows; the labels are looked up on the grid
            run.label_rows = rows_labelled
            run.rows = pk.inv[rows_labelled]
        self._refresh_if_stale(need_grad)  # bf16 operands / composed adapter rows of the trainable parameters
        use_ans = bool(m.n_ans) and not mlm
        if logit_rows is not None:
            if need_grad or full_labels is not None:
                raise RuntimeError("logit_rows is an inference-time option (no labels, no gradient bookkeeping)")
            run.logit_rows = logit_rows.to(self.dev).to(torch.int32).contiguous().view(-1)
            if pk is not None:
                run.logit_rows = pk.inv[run.logit_rows.long()].to(torch.int32)
        with L.seed_word(run.seed_word):
            logits, loss_t = self._forward(run, input_ids.contiguous(), video, use_ans, want_hidden or want_attn)
        Vout = self.n_ans if use_ans else self.V
        if logit_rows is not None:
            res = {"logits": logits[:, :Vout], "loss": None, "run": run}
            if want_hidden:
                res["hidden_states"] = run.hidden_out
            if want_attn:
                res["attentions"] = tuple(run.attn_out)
            return res
        res = {"logits": logits.view(B, S, -1)[:, :, :Vout] if logits is not None else None, "loss": None, "run": run}
        # (packed rows: `logits` is the [B*S, V] grid tensor here too -- allocated in _forward, filled on access)
        if want_hidden:
            res["hidden_states"] = run.hidden_out
        if want_attn:
            res["attentions"] = tuple(run.attn_out)
        if need_grad and res["logits"] is not None:
            # one autograd node for the whole model; both outputs are differentiable: the loss (MLM training,
            # main.py:67-84) and the logits (downstream fine-tuning computes its own loss on them, videoqa.py:66-83,
            # mc.py:64-92)
            lt = loss_t if loss_t is not None else torch.zeros((), dtype=F32, device=self.dev)
            loss_o, logits_o = _StepFn.apply(self, run, lt, res["logits"], *[self.named[n] for n in self.order])
            res["logits"] = logits_o
            res["loss"] = loss_o if full_labels is not None else None
        elif full_labels is not None:
            res["loss"] = loss_t
        return res

    def _make_packing(self, mask, full_labels, logit_rows, B, S, T) -> Optional[Packing]:
        """Packing of this batch, or None when no row can be dropped (one host read of B lengths)."""
        dev = self.dev
        keep = mask.view(B, S) != 0
        if full_labels is not None:
            keep = keep | (full_labels.view(B, S) != -100)
        if logit_rows is not None:
            want = torch.zeros(B * S, dtype=torch.bool, device=dev)
            want[logit_rows.to(dev).long().view(-1)] = True
            keep = keep | want.view(B, S)
        pos1 = torch.arange(1, S + 1, device=dev, dtype=torch.int32)
        plen_h = [max(int(v), T, 1) for v in (keep.to(torch.int32) * pos1).amax(1).tolist()]
        n = sum(plen_h)
        if n >= B * S:
            return None
        offs = [0]
        for v in plen_h:
            offs.append(offs[-1] + v)
        row0 = torch.tensor(offs, dtype=torch.int32, device=dev)
        b_of = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(plen_h, device=dev), output_size=n)
        pos = torch.arange(n, device=dev) - row0[:-1].long()[b_of]
        sel = b_of * S + pos
        inv = torch.full((B * S,), -1, dtype=torch.int64, device=dev)
        inv[sel] = torch.arange(n, device=dev)
        return Packing(row0=row0, sel=sel, inv=inv, pos=pos, n=n)

    # ------------------------------------------------------------------ forward
    def _ln(self, run, name, *, y, resid: Optional[Stream], N, p_drop=0.0, rowmask=None, want_f32=False, tail=0):
        H = self.H
        g, b = self.P[name + ".weight"], self.P[name + ".bias"]
        t = torch.empty(N, H, dtype=F32, device=self.dev)
        stats = torch.empty(N, 2, dtype=F32, device=self.dev)
        full = torch.empty(N + tail, H, dtype=BF16, device=self.dev)
        ob = full[:N]
        of = torch.empty(N, H, dtype=F32, device=self.dev) if want_f32 else None
        seed = run.next_seed() if p_drop > 0 else 0
        r_norm = resid.norm.as_args() if resid is not None and resid.norm is not None else None
        r_plain = resid.plain if resid is not None and r_norm is None else None
        L.ln_fwd(y=y, p_drop=p_drop, seed=seed, r_plain=r_plain, r_norm=r_norm,
                 gamma=g, beta=b, eps=self.cfg.layer_norm_eps, rowmask=rowmask, out_t=t, out_stats=stats, out_bf16=ob,
                 out_f32=of, N=N, H=H)
        return Stream(bf16=ob, norm=NormRef(t, stats, g, b, rowmask), plain=of, full=full if tail else None), seed

    def _adapter_fwd(self, run, ent, x_f32, x_bf16, N, z=None, seed=0):
        """y = x + up(drop(relu(down(x))))  (model/adapter.py:33-45).  The bottleneck z either comes from the merged
        dense + down-projection GEMM of the caller (z given) or from fbl_adapter_down_fwd; the up-projection is a GEMM
        with bias + residual epilogue."""
        A, Ap = ent["A"], ent["Ap"]
        if z is None:
            z = torch.zeros(N, Ap, dtype=BF16, device=self.dev) if Ap != A else torch.empty(N, Ap, dtype=BF16, device=self.dev)
            seed = run.next_seed() if run.p_ad > 0 else 0
            L.adapter_down_fwd(x_bf16, ent["down"], ent["bd"], z, A=A, p_drop=run.p_ad, seed=seed)  # ReLU + dropout in the epilogue
        y = torch.empty(N, self.H, dtype=F32, device=self.dev)
        L.gemm(z, ent["up"], bias=ent["bu"], aux=x_f32, aux_kind=L.AUX_ADD_F32, out_f32=y)
        return y, z, seed

    def _dense_adapter(self, run, li, x_bf16, W, wkey, bkey, ent, A, merged, N):
        """dense -> adapter (model/deberta.py:255-257 / :329-331): returns (y fp32 = adapter output, dense output bf16, z,
        dropout seed).  Merged form: one GEMM gives the dense output and the adapter bottleneck."""
        H, dev = self.H, self.dev
        o32 = torch.empty(N, H, dtype=F32, device=dev)
        # the bf16 copy of the dense output is the adapter's GEMM operand: with the merged GEMM (the bottleneck comes out
        # of the same launch) only the backward reads it -- an inference forward does not write it (26 MB per site)
        need_ob = run.save or ent is None or not (merged and ent["Ap"] == A)
        ob = torch.empty(N, H, dtype=BF16, device=dev) if need_ob else None
        if ent is None:
            L.gemm(x_bf16, W[wkey], bias=W[bkey], out_f32=o32, out_bf16=ob)
            return o32, ob, None, 0
        if merged and ent["Ap"] == A:
            ev = getattr(self, "_compose_ev", None)
            if ev is not None:  # the composed rows of this layer must be there (layer 0: first event, others: second)
                torch.cuda.current_stream().wait_event(ev[0] if li == 0 else ev[1])
            z = torch.empty(N, A, dtype=BF16, device=dev)
            seed = run.next_seed() if run.p_ad > 0 else 0
            L.dense_adapter_down_fwd(x_bf16, W[wkey + "M"], W[bkey + "M"], H, z, y_f32=o32, y_bf16=ob, p_drop=run.p_ad, seed=seed)
            y, z, seed = self._adapter_fwd(run, ent, o32, ob, N, z=z, seed=seed)
        else:
            L.gemm(x_bf16, W[wkey], bias=W[bkey], out_f32=o32, out_bf16=ob)
            y, z, seed = self._adapter_fwd(run, ent, o32, ob, N)
        return y, ob, z, seed

    def _dense_adapter_ln(self, run, li, x_bf16, W, wkey, bkey, ent, A, merged, N, ln_name, resid: Stream, tail=0):
        """dense -> adapter -> dropout -> LayerNorm(. + resid)  (model/deberta.py:254-260 / 328-334).  Returns (output
        Stream, dense output bf16, z, adapter seed, LayerNorm-dropout seed).  With a merged adapter site the block is three
        launches -- merged dense + down-projection GEMM (bf16 y, z), up-projection GEMM whose epilogue adds the adapter
        input, applies the block's dropout and adds the residual (writes the pre-norm tensor t once), LayerNorm statistics
        + bf16 operand -- and neither the fp32 dense output nor the fp32 adapter output exist in HBM."""
        H, dev = self.H, self.dev
        if not (self.fuse_tail and ent is not None and merged and ent["Ap"] == A):
            y, ob, z, seed_ad = self._dense_adapter(run, li, x_bf16, W, wkey, bkey, ent, A, merged, N)
            out, seed_ln = self._ln(run, ln_name, y=y, resid=resid, N=N, p_drop=run.p_hid, tail=tail)
            return out, ob, z, seed_ad, seed_ln
        ev = getattr(self, "_compose_ev", None)
        if ev is not None:  # the composed rows of this layer must be there (layer 0: first event, others: second)
            torch.cuda.current_stream().wait_event(ev[0] if li == 0 else ev[1])
        ob = torch.empty(N, H, dtype=BF16, device=dev)
        z = torch.empty(N, A, dtype=BF16, device=dev)
        seed_ad = run.next_seed() if run.p_ad > 0 else 0
        L.dense_adapter_down_fwd(x_bf16, W[wkey + "M"], W[bkey + "M"], H, z, y_bf16=ob, p_drop=run.p_ad, seed=seed_ad)
        g, b = self.P[ln_name + ".weight"], self.P[ln_name + ".bias"]
        t = torch.empty(N, H, dtype=F32, device=dev)
        stats = torch.empty(N, 2, dtype=F32, device=dev)
        full = torch.empty(N + tail, H, dtype=BF16, device=dev)
        seed_ln = run.next_seed() if run.p_hid > 0 else 0
        r_norm = resid.norm.as_args() if resid.norm is not None else None
        L.adapter_up_resid_fwd(z, ent["up"], ent["bu"], ob, t, A=A, p_drop=run.p_hid, seed=seed_ln,
                               r_plain=resid.plain if r_norm is None else None, r_norm=r_norm)
        L.ln_fwd(y=t, gamma=g, beta=b, eps=self.cfg.layer_norm_eps, out_stats=stats, out_bf16=full[:N], N=N, H=H)
        return (Stream(bf16=full[:N], norm=NormRef(t, stats, g, b, None), full=full if tail else None), ob, z, seed_ad,
                seed_ln)

    def _pos_proj_cached(self, li: int, W, Rb: torch.Tensor) -> torch.Tensor:
        """[PQ|PK] = R.[Wq;Wk]^T + b of layer `li` ([2*span, 2H] bf16, model/deberta.py:847-853) for inference forwards inside
        model.weights_frozen(): R = LayerNorm_enc(rel_embeddings) only changes with the trainable parameters, and every
        rebuild of their operands (refresh_trainable_operands) drops this cache."""
        c = self._pos_cache.get(li)
        if c is None:
            c = torch.empty(self.span2, 2 * self.H, dtype=BF16, device=self.dev)
            L.gemm(Rb, W["Wqkv"][: 2 * self.H], bias=W["bqkv"][: 2 * self.H], out_bf16=c)
            self._pos_cache[li] = c
        return c

    def _layer_fwd(self, run, li: int, kv: Stream, q: Optional[Stream], Rb: torch.Tensor):
        """One execution of encoder layer ``li`` (model/deberta.py:351-375); q != None is the EMD form where the
        query stream (and the attention residual, :290-292) differs from the key/value stream."""
        W = self.Lw[li]
        B, S, H, I, nh = run.B, run.S, self.H, self.I, self.nh
        N = run.N
        Sp = _ru(S, 64)
        dev = self.dev
        sv = LayerSave(li=li, emd=q is not None)
        # One GEMM gives Q|K|V for the N tokens AND the shared-key position projections [PQ|PK] = R.[Wq|Wk]^T (:847-853):
        # the relative-position table R (pos_dropout applied, :779) is written into the P tail rows of the activations.
        P_ = self.span2
        if run.p_hid > 0:
            sv.seed_pos = run.next_seed()

        def put_r(full):
            if run.p_hid > 0:
                L.dropout_f32(run.R32, run.p_hid, sv.seed_pos, out_bf16=full[N:])
            else:
                full[N:].copy_(Rb)

        # Inference shortcuts (no gradient bookkeeping, no position dropout -- results are bit-identical to the general path):
        #  * inside model.weights_frozen() the position projections [PQ|PK] of a layer are the same in every forward: computed
        #    once per layer and kept (dropped whenever the trainable operands are rebuilt), the QKV GEMM then has N rows;
        #  * the second pass of the enhanced mask decoder projects the SAME key/value stream with the same weights as the
        #    first (model/deberta.py:1395-1408): only its query projection is new, written over the first pass's Q columns.
        infer = not run.save and run.p_hid == 0
        pqk_c = (self._pos_proj_cached(li, W, Rb)
                 if infer and getattr(self.m, "_weights_frozen", 0) > 0 and not self._no_pos_cache else None)
        reuse = infer and q is not None and getattr(run, "emd_qkv", None) is not None
        if reuse:
            qkv = run.emd_qkv
            if pqk_c is None:
                put_r(q.full)
            L.gemm(q.full if pqk_c is None else q.bf16, W["Wqkv"][:H], bias=W["bqkv"][:H], out_bf16=qkv[:, :H])
        else:
            rows_in = N if pqk_c is not None else N + P_
            qkv = torch.empty(rows_in, 3 * H, dtype=BF16, device=dev)
            kin = kv.full[:rows_in]
            if pqk_c is None:
                put_r(kv.full)
            if q is None:
                L.gemm(kin, W["Wqkv"], bias=W["bqkv"], out_bf16=qkv)
            else:
                if pqk_c is None:
                    put_r(q.full)
                L.gemm(q.full[:rows_in], W["Wqkv"][:H], bias=W["bqkv"][:H], out_bf16=qkv[:, :H])
                L.gemm(kin, W["Wqkv"][H:], bias=W["bqkv"][H:], out_bf16=qkv[:, H:])
                if infer:
                    run.emd_qkv = qkv
        if pqk_c is not None:
            pq, pk = pqk_c[:, :H], pqk_c[:, H:]
        else:
            pq, pk = qkv[N:, :H], qkv[N:, H:2 * H]
        ctx = torch.empty(N, H, dtype=BF16, device=dev)
        lse = torch.empty(B, nh, S, dtype=F32, device=dev)
        sv.seed_att = run.next_seed() if run.p_att > 0 else 0
        # training: the forward leaves its un-normalised probabilities (bf16, 157 MB per execution at the bench shape; only the
        # tile pairs it visits are written) so that the backward does not recompute the scores (attn_bwd.hip, kernel A / dsp)
        psave = msave = None
        if run.save and self.attn_save_p:
            psave = torch.empty(B, nh, Sp, Sp, dtype=BF16, device=dev)
            msave = torch.empty(B, nh, Sp // 64, S, dtype=F32, device=dev)
        L.disent_attn_fwd(qkv[:N, :H], qkv[:N, H:2 * H], qkv[:N, 2 * H:], pk, pq, self.relidx(S), run.mask_i32,
                          1.0 / math.sqrt(64 * 3), ctx, lse, B, S, Sp, nh, self.span2, p_drop=run.p_att,
                          seed=sv.seed_att, klen=run.klen, border=run.border, lin=self.lin_span,
                          row0=run.pk.row0 if run.pk is not None else None, psave=psave, msave=msave)
        sv.psave, sv.msave = psave, msave
        if getattr(run, "want_attn", False) and q is None:
            # output_attentions=True: the encoder layers' probabilities (model/deberta.py:544-560; the enhanced-mask-decoder
            # passes are called with return_att=False, :1395-1408), materialised by a plain kernel from the stored lse
            probs = torch.empty(B, nh, S, S, dtype=F32, device=dev)
            L.disent_attn_probs(qkv[:N, :H], qkv[:N, H:2 * H], pk, pq, self.relidx(S), run.mask_i32, lse,
                                1.0 / math.sqrt(64 * 3), probs, B, S, nh)
            run.attn_out.append(probs)
        # attention output: dense -> adapter -> dropout -> LN(. + residual)   (:254-260)
        ad = self.ad[li]
        p = f"deberta.encoder.layer.{li}"
        a, ob, z1, sv.seed_ad1, sv.seed_ln1 = self._dense_adapter_ln(
            run, li, ctx, W, "Wo", "bo", ad.get("a1"), self.A1, self.merge1, N, p + ".attention.output.LayerNorm",
            resid=(q if q is not None else kv))
        # FFN: gelu(dense) -> dense -> adapter -> dropout -> LN(. + a)          (:310-313, :328-334)
        h = torch.empty(N, I, dtype=BF16, device=dev)
        hpre = torch.empty(N, I, dtype=BF16, device=dev) if run.save else None
        # training: the epilogue stores gelu'(pre) (bf16) next to gelu(pre) so the backward epilogue is a plain multiply
        L.gemm(a.bf16, W["Wi"], bias=W["bi"], act=L.ACT_GELU_GRAD if run.save else L.ACT_GELU, out_bf16=h, out_pre=hpre)
        out, fb, z2, sv.seed_ad2, sv.seed_ln2 = self._dense_adapter_ln(
            run, li, h, W, "Wd", "bd", ad.get("a2"), self.A2, self.merge2, N, p + ".output.LayerNorm",
            resid=Stream(bf16=a.bf16, norm=a.norm), tail=self.span2)
        if run.save:
            sv.qkv, sv.pqk, sv.ctx, sv.lse = qkv[:N], qkv[N:, : 2 * H], ctx, lse  # (run.save: never the inference shortcuts)
            sv.ob, sv.z1, sv.ln1 = ob, z1, a.norm
            sv.hpre, sv.fb, sv.z2, sv.ln2 = hpre, fb, z2, out.norm
            run.layers.append(sv)
        return out

    def _forward(self, run, input_ids, video, use_ans, want_hidden):
        cfg, H, dev = self.cfg, self.H, self.dev
        B, S, T, Lt = run.B, run.S, run.T, run.Lt
        pk = getattr(run, "pk", None)
        if not run.N:
            run.N = B * S
        N = run.N
        run.mask_i32 = run.mask  # [B*S]: the attention kernels index the mask on the padded grid
        run.rowmask = run.mask if pk is None else run.mask[pk.sel].contiguous()  # per activation row
        pos1 = torch.arange(1, S + 1, device=dev, dtype=torch.int32)
        run.klen = (run.mask.view(B, S) * pos1).amax(1).to(torch.int32).contiguous()  # last valid position + 1
        # attention work per sample grows with klen^2: the attention kernels dispatch the samples longest first
        run.border = torch.argsort(run.klen, descending=True, stable=True).to(torch.int32).contiguous()
        mask_f = run.rowmask.to(F32)
        run.mask_f = mask_f
        # ---- embeddings (model/deberta.py:997-1058): cat(linear_video(video), E[ids]) -> LN -> *mask -> dropout
        vproj = None
        if T:
            vb = torch.zeros(B * T, self.Fp, dtype=BF16, device=dev)
            vb[:, : self.F] = video.reshape(B * T, self.F)
            vproj = torch.empty(B * T, H, dtype=F32, device=dev)
            L.gemm(vb, self.Wv, bias=self.P["deberta.embeddings.linear_video.bias"], out_f32=vproj)
            run.video_bf16 = vb
        t0 = torch.empty(B * S, H, dtype=F32, device=dev)
        L.embed_gather(input_ids, self.E32, vproj, T, t0)
        if pk is not None:
            t0 = t0.index_select(0, pk.sel)
        want_plain = run.p_hid > 0
        emb, _ = self._ln(run, "deberta.embeddings.LayerNorm", y=t0, resid=None, N=N, rowmask=run.rowmask,
                          want_f32=want_plain, tail=self.span2)
        run.emb_norm = emb.norm
        if want_plain:  # post-LN dropout: materialise x0
            run.seed_emb = run.next_seed()
            L.dropout_f32(emb.plain, run.p_hid, run.seed_emb, out_f32=emb.plain, out_bf16=emb.bf16)
            emb = Stream(bf16=emb.bf16, plain=emb.plain, full=emb.full)
        # ---- relative-position table: R = LayerNorm_enc(rel_embeddings.weight)   (:474-478)
        r, _ = self._ln(run, "deberta.encoder.LayerNorm", y=self.rel_emb, resid=None, N=self.span2, want_f32=run.p_hid > 0)
        run.rel_norm, run.R32, Rb = r.norm, r.plain, r.bf16
        # ---- encoder (:507-575)
        hs: List[Stream] = [emb]
        x = emb
        nL = self.nL
        for i in range(nL):
            if i == nL - 1 and self.skip_dead_layer and not want_hidden:
                break  # its output feeds nothing; executed only when hidden_states are requested
            if i == nL - 1:
                keep, run.save = run.save, False
                x = self._layer_fwd(run, i, x, None, Rb)
                run.save = keep
            else:
                x = self._layer_fwd(run, i, x, None, Rb)
            if i == 0 and cfg.conv_kernel_size:
                x = self._conv_fwd(run, emb, x)
            hs.append(x)
        # ---- enhanced mask decoder (:1382-1412): q0 = pos_emb + hs[-2]; two passes of the last layer
        kv = hs[nL - 1]
        q32 = torch.empty(N, H, dtype=F32, device=dev)
        qfull = torch.empty(N + self.span2, H, dtype=BF16, device=dev)
        if pk is None:
            self._materialize(kv, add=self.pos_emb, S=S, out_f32=q32, out_bf16=qfull[:N])
        else:  # packed rows: the position of a row is not row % S (once per step: plain tensor ops, same fp32 arithmetic)
            n_ = kv.norm
            if n_ is not None:
                L.ln_materialize(n_.t, n_.stats, n_.gamma, n_.beta, rowmask=n_.rowmask, out_f32=q32)
            else:
                q32.copy_(kv.plain)
            q32.add_(self.pos_emb.index_select(0, pk.pos))
            qfull[:N].copy_(q32)
        q = Stream(bf16=qfull[:N], plain=q32, full=qfull)
        for _ in range(2):
            q = self._layer_fwd(run, nL - 1, kv, q, Rb)
        if want_hidden:
            if pk is None:
                run.hidden_out = tuple(self._materialize(s).view(B, S, H) for s in hs)
            else:  # (positions without a row read as zero)
                run.hidden_out = tuple(torch.zeros(B * S, H, dtype=F32, device=dev).index_copy_(0, pk.sel, self._materialize(s))
                                       .view(B, S, H) for s in hs)
        # ---- MLM head (:1544-1558): LN(gelu(dense(x))) . table^T + bias
        hin = q.bf16
        rows_only = getattr(run, "logit_rows", None)
        # an inference forward that is only asked for the loss (evaluate, main.py:139) runs the head's dense + GELU +
        # LayerNorm on the labelled rows only, like the vocabulary GEMM below; fill_logits redoes it on every row if the
        # logits are read after all
        loss_rows = (rows_only is None and run.labels is not None and not run.save and not self.eager_logits
                     and 0 < run.rows.numel() < N)
        n_all = B * S
        if loss_rows:
            run.head_in_all = q.bf16
            rows_only = run.rows.to(torch.int32)
        if rows_only is not None:  # inference on selected token rows (the [MASK] rows of videoqa.py:164-168 / mc.py:166-170)
            N = rows_only.numel()
            hin = torch.empty(N, H, dtype=BF16, device=dev)
            if N:
                L.gather_rows_bf16(q.bf16, rows_only, hin)
        hp, hl = self._head_stage(run, hin, N)
        run.head_pre, run.head_norm, run.head_ln_bf16 = hp, hl.norm, hl.bf16
        if use_ans:
            Vout, table, bias = self.n_ans, self.Ansb, self.ans_bias
        else:
            Vout, table, bias = self.V, self.Eb, self.head_bias
        ldv = _ru(Vout, 64)
        run.Vout = Vout
        run.head_table, run.head_bias, run.ldv = table, bias, ldv
        if run.labels is not None:
            # A loss is asked for: it only needs the labelled rows.  Their logits come from a small GEMM so CE (and the
            # backward) can follow at once; the full [N, V] logits tensor -- an OUTPUT of the reference API that neither
            # main.py's train_one_epoch nor evaluate reads (main.py:67,139) -- is allocated but only FILLED on first
            # access (MaskedLMOutput -> fill_logits): 4.4 GB of writes and 3.4 TFLOP per step at the xlarge vocabulary.
            rows = run.rows
            run.rows_i32 = rows.to(torch.int32)
            R = rows.numel()
            run.loss_acc = L.zeros(2, dtype=F32, device=dev)
            if R > 0:
                if loss_rows:
                    hrows = hl.bf16  # the head already ran on exactly these rows
                else:
                    hrows = torch.empty(R, H, dtype=BF16, device=dev)
                    L.gather_rows_bf16(hl.bf16, run.rows_i32, hrows)
                lc = torch.empty(R, ldv, dtype=F32, device=dev)
                L.gemm(hrows, table, bias=bias, out_f32=lc, N=Vout)
                lab_rows = getattr(run, "label_rows", None)  # (packed rows: `rows` are activation rows, the labels live on the grid)
                run.labels_c = run.labels[rows if lab_rows is None else lab_rows].contiguous()
                run.row_lse = torch.empty(R, dtype=F32, device=dev)
                L.ce_fwd(lc, run.labels_c, Vout, run.row_lse, run.loss_acc)
                run.logits_c = lc
            loss_t = run.loss_acc[0] / run.loss_acc[1]  # mean over labelled rows (CrossEntropyLoss, :1483-1488)
            run.logits = torch.empty(n_all, ldv, dtype=F32, device=dev)
            run.logits_pending = True
            if self.eager_logits:
                self.fill_logits(run)
            return run.logits, loss_t
        logits = torch.empty(N, ldv, dtype=F32, device=dev)
        L.gemm(hl.bf16, table, bias=bias, out_f32=logits, N=Vout)
        run.logits = logits
        return logits, None

    def fill_logits(self, run):
        """Full [N, V] logits of a forward that only computed the labelled rows (see _forward); idempotent.  Writes into
        the storage of the tensor already handed out (and already wired into the autograd node), on the current stream."""
        if getattr(run, "logits_pending", False):
            run.logits_pending = False
            hin_all = getattr(run, "head_in_all", None)
            if hin_all is not None:  # the forward ran the head on the labelled rows only: now on every row
                _, hl = self._head_stage(run, hin_all, hin_all.shape[0])
                run.head_ln_bf16, run.head_in_all = hl.bf16, None
            pk = getattr(run, "pk", None)
            if pk is None:
                L.gemm(run.head_ln_bf16, run.head_table, bias=run.head_bias, out_f32=run.logits, N=run.Vout)
            else:  # packed rows: grid positions without a row read as zero; the others arrive in slabs of 1024 rows
                run.logits.zero_()
                for r0 in range(0, pk.n, 1024):
                    r1 = min(pk.n, r0 + 1024)
                    slab = torch.empty(r1 - r0, run.logits.shape[1], dtype=F32, device=self.dev)
                    L.gemm(run.head_ln_bf16[r0:r1], run.head_table, bias=run.head_bias, out_f32=slab, N=run.Vout)
                    run.logits.index_copy_(0, pk.sel[r0:r1], slab)

    def _head_stage(self, run, hin, N):
        """prediction head in front of the vocabulary GEMM (model/deberta.py:1544-1552): LayerNorm(gelu(dense(x)))"""
        hp = torch.empty(N, self.H, dtype=F32, device=self.dev)
        L.gemm(hin, self.Wh, bias=self.bh, out_f32=hp)
        hg = torch.empty(N, self.H, dtype=F32, device=self.dev)
        L.dropout_gelu_fwd(hp, 0.0, 0, hg)
        hl, _ = self._ln(run, "lm_predictions.lm_head.LayerNorm", y=hg, resid=None, N=N)
        return hp, hl

    def _materialize(self, s: Stream, add=None, S=1, out_f32=None, out_bf16=None):
        N, H = s.bf16.shape
        if s.plain is not None and add is None and out_bf16 is None:
            return s.plain
        if out_f32 is None:
            out_f32 = torch.empty(N, H, dtype=F32, device=self.dev)
        if s.norm is not None:
            n = s.norm
            L.ln_materialize(n.t, n.stats, n.gamma, n.beta, rowmask=n.rowmask, add_bcast=add, S=S, out_f32=out_f32,
                             out_bf16=out_bf16)
        else:
            v = s.plain if add is None else (s.plain.view(-1, S, H) + add[:S]).view(N, H)
            out_f32.copy_(v)
            if out_bf16 is not None:
                out_bf16.copy_(v)
        return out_f32

    def _conv_fwd(self, run, emb: Stream, l0: Stream) -> Stream:
        """ConvLayer (:395-419): LN(l0 + gelu(drop(mask * conv1d_k3(emb)))) * mask, conv as a K=3H GEMM on an im2col."""
        B, S, H, dev = run.B, run.S, self.H, self.dev
        N = run.N
        pk = getattr(run, "pk", None)
        if pk is None:
            col = torch.empty(N, 3 * H, dtype=BF16, device=dev)
            L.im2col3(emb.bf16, col, B, S, H)
        else:  # packed rows: the neighbours of a position are found on the padded grid (masked embedding rows are zero there too)
            grid = torch.zeros(B * S, H, dtype=BF16, device=dev).index_copy_(0, pk.sel, emb.bf16)
            colg = torch.empty(B * S, 3 * H, dtype=BF16, device=dev)
            L.im2col3(grid, colg, B, S, H)
            col = colg.index_select(0, pk.sel)
        c = torch.empty(N, H, dtype=F32, device=dev)
        L.gemm(col, self.Wc, bias=self.bc, rowscale=run.mask_f, out_f32=c)
        y = torch.empty(N, H, dtype=F32, device=dev)
        run.seed_conv = run.next_seed() if run.p_hid > 0 else 0
        L.dropout_gelu_fwd(c, run.p_hid, run.seed_conv, y)
        out, _ = self._ln(run, "deberta.encoder.conv.LayerNorm", y=y, resid=l0, N=N, rowmask=run.rowmask, tail=self.span2)
        run.conv_c, run.conv_norm = c, out.norm
        return out

    # ------------------------------------------------------------------ backward
    def _ln_bwd(self, name, dout, norm: NormRef, p_drop, seed, want_dy_bf16=True, dysum=None, tail=0, dy_out=None):
        """dysum: optional [H] accumulator for colsum(dy) = the bias gradient of whatever produced y (adapter up.bias).
        tail > 0: dt is the first N rows of a [N+tail, H] buffer whose tail is zeroed (aux operand of the dX GEMM that
        also produces the position-table gradient rows); the full buffer is returned as third value."""
        N, H = dout.shape
        dt_full = torch.empty(N + tail, H, dtype=F32, device=self.dev)
        dt = dt_full[:N]
        if tail:
            dt_full[N:].zero_()
        dyb = dy_out if dy_out is not None else (torch.empty(N, H, dtype=BF16, device=self.dev) if want_dy_bf16 else None)
        L.ln_bwd(dout, norm.t, norm.stats, norm.gamma, rowmask=norm.rowmask, p_drop=p_drop, seed=seed, out_dt=dt,
                 out_dy_bf16=dyb, dgamma=self.G.get(name + ".weight"), dbeta=self.G.get(name + ".bias"), dysum=dysum,
                 ws=self._ln_ws)
        if tail:
            return dt, dyb, dt_full
        return dt, dyb

    def _adapter_bwd(self, run, ent, dyb, z, xin_b, seed, dz_out=None, pooled=None):
        """Backward of _adapter_fwd.  dyb: grad of the adapter output (bf16 [N,H]); returns grad of its input (bf16).
        dz_out: the caller folds dx = dy + dz.Wd into the next GEMM ([dy | dz] operand, _layer_bwd): dz is written there
        (a column slice of that operand), no dx is formed and None is returned."""
        N, H = dyb.shape
        A, Ap = ent["A"], ent["Ap"]
        dev = self.dev
        upT, downT = self._adapter_bwd_operands(ent)
        if dz_out is not None:
            dz = dz_out
        else:
            dz = torch.zeros(N, Ap, dtype=BF16, device=dev) if Ap != A else torch.empty(N, Ap, dtype=BF16, device=dev)
        inv_keep = 1.0 / (1.0 - run.p_ad) if run.p_ad > 0 else 1.0
        L.gemm(dyb, upT, alpha=inv_keep, aux=z, aux_kind=L.AUX_MUL_POS_BF16, out_bf16=dz, N=A)
        dx = None
        if dz_out is None:
            dx = torch.empty(N, H, dtype=BF16, device=dev)
            L.gemm(dz, downT, aux=dyb, aux_kind=L.AUX_ADD_BF16, out_bf16=dx)
        sk = max(2, min(16, N // 512))
        nm = ent["name"]

        if Ap <= 256 and H % 8 == 0 and not self.dw_on_side:
            # dWu += dy^T z, dWd += dz^T x, dbd += colsum(dz) are NOT launched here: one adapter alone has 48 output tiles.
            # The operands are parked until `dw_group` adapters are pending and go through ONE launch
            # (fbl_adapter_bwd_dw: every tile walks all rows, no split-K round trip) on the MAIN stream -- its workgroups live
            # for the whole pass, and a resident side-stream workgroup keeps the one-per-CU tiles of the big GEMMs off its CU.
            # (up.bias's gradient colsum(dy) comes out of ln_bwd.)
            run.dw_pending.append((A, nm, (dyb, z, dz, xin_b), run.dw_count, pooled))
            run.dw_count += 1
            return dx

        def dw_work(ws, cs_ws):  # generic route, launched right away: bottlenecks wider than 256 (or engine_options dw_on_side)
            L.gemm_tn_acc(dyb, z, self.G[nm + ".up.weight"], ws, N=A, splitk=sk)      # dWu[H,A] += dy^T z
            L.gemm_tn_acc(dz, xin_b, self.G[nm + ".down.weight"], ws, M=A, splitk=sk)  # dWd[A,H] += dz^T x
            L.colsum(dz, self.G[nm + ".down.bias"], cs_ws, cols=A)

        if self.use_side_stream:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)  # inputs (dyb, z, dz, xin_b) are ready once main reaches this point
            with torch.cuda.stream(self.side):
                dw_work(self.side_ws, self.side_cs_ws)
            if not torch.cuda.is_current_stream_capturing():
                for t in (dyb, z, dz, xin_b):
                    t.record_stream(self.side)  # keep the allocator from recycling them before the side stream is done
            else:  # (inside a capture: referenced for the life of the captured step instead, see _pos_grad_async)
                run.__dict__.setdefault("_pos_keep", []).extend((dyb, z, dz, xin_b))
            run.side_used = True
            if self.reducer is not None:  # what a data-parallel bucket has to wait for: the dW work queued so far
                run.dw_event = torch.cuda.Event()
                run.dw_event.record(self.side)
        else:
            dw_work(self.sk_ws, self._cs_ws)
        return dx

    def _dw_flush(self, run, red=None, force=False):
        """Launch the parked adapter-gradient products (see _adapter_bwd) once `dw_group` distinct adapters are pending -- one
        launch of 2 x H/64 workgroups per adapter; 16 adapters = 768 workgroups = three per CU, all resident -- or all of
        them (force), and only then tell the gradient reducer about the stages whose buckets they complete.  An adapter
        appears once per launch: the second execution of the last layer (enhanced mask decoder) waits for the next one,
        so that no launch has a double-length tile."""
        pend = run.dw_pending
        while pend and (force or len({rec[1] for rec in pend}) >= self.dw_group):
            take, names, rest = {}, set(), []
            for rec in pend:
                A, nm, seg = rec[:3]
                if nm in names or len(names) >= min(self.dw_group, L.ADW_MAX_ADAPTERS):
                    rest.append(rec)
                else:
                    names.add(nm)
                    take.setdefault(A, []).append((nm, seg))
            for A, recs in take.items():
                L.adapter_bwd_dw([([seg], self.G[nm + ".up.weight"], self.G[nm + ".down.weight"], self.G[nm + ".down.bias"])
                                  for nm, seg in recs], A=A)
            kept = {id(r) for r in rest}
            for rec in pend:  # [dy | dz | 0] operands whose last reader has now been enqueued go back to the pool (same stream)
                if id(rec) not in kept and rec[4] is not None and tuple(rec[4].shape) == self._dyz_shape:
                    self._dyz_pool.setdefault(self._dyz_shape, []).append(rec[4])  # (buffers of an earlier batch shape are dropped)
            pend[:] = rest
        if red is not None:  # a finished stage is final once none of ITS adapters has a parked product left (any order: the
            # repeated last layer waits one launch longer than the layers behind it, and must not hold their buckets back)
            parked = {self._bucket_key(rec[1] + ".up.weight") for rec in pend}
            still = []
            for key in run.dw_ready_keys:
                if key in parked:
                    still.append(key)
                else:
                    red.ready(key)
            run.dw_ready_keys[:] = still

    def _layer_bwd(self, run, sv: "LayerSave", dout: torch.Tensor):
        """Backward of one layer execution.  dout: fp32 grad of its output.  Returns (dx, None) for ordinary encoder layers (one
        input stream) and (dq_in fp32, [dK | dV] bf16 operand of the key/value-stream gradient) for the decoder form."""
        li = sv.li
        W, ad = self.Lw[li], self.ad[li]
        H, I, dev = self.H, self.I, self.dev
        N = dout.shape[0]
        p = f"deberta.encoder.layer.{li}"
        dt2, dy2 = self._ln_bwd(p + ".output.LayerNorm", dout, sv.ln2, run.p_hid, sv.seed_ln2,
                                dysum=self.G[ad["a2"]["name"] + ".up.bias"] if "a2" in ad else None)
        df = dy2
        if "a2" in ad:
            df = self._adapter_bwd(run, ad["a2"], dy2, sv.z2, sv.fb, sv.seed_ad2)
        dh = torch.empty(N, I, dtype=BF16, device=dev)
        L.gemm(df, W["WdT"], aux=sv.hpre, aux_kind=L.AUX_MUL_BF16, out_bf16=dh)  # sv.hpre holds gelu'(pre)
        da = torch.empty(N, H, dtype=F32, device=dev)
        L.gemm(dh, W["WiT"], aux=dt2, aux_kind=L.AUX_ADD_F32, out_f32=da)
        del dh
        dctx = torch.empty(N, H, dtype=BF16, device=dev)
        fold = "a1" in ad and self.fold_dx and "WoF" in W and ad["a1"]["Ap"] == ad["a1"]["A"]
        if fold:
            # dctx = dx . Wo with dx = dy + dz . Wd  ==  [dy | dz] . [Wo^T | (Wd.Wo)^T]^T: LayerNorm backward and the dz GEMM
            # write the two column blocks of ONE operand and the dense dX GEMM runs with K = H + A (+ zero padding) -- the
            # K = A GEMM that formed dx, its 26 MB output and its re-read are gone (25 us per layer execution)
            A1 = ad["a1"]["A"]
            # The K padding columns only have to hold finite values (their weight columns are zero): buffers come from a small
            # pool whose padding was zeroed ONCE -- nobody writes those columns -- instead of a strided fill per layer execution
            # (25 launches a step).  A buffer returns to the pool when the parked gradient products that read its [dy | dz]
            # blocks have been enqueued (_dw_flush).  Captured steps and the immediate-launch route allocate as before.
            pooled = None
            can_pool = (not torch.cuda.is_current_stream_capturing() and ad["a1"]["Ap"] <= 256 and H % 8 == 0 and not self.dw_on_side)
            if can_pool and self._dyz_shape != (N, self.Kf1):  # a new batch shape (text padded to the longest sample): one pool, of
                self._dyz_pool.clear()                           # the current shape only -- never one per shape seen
                self._dyz_shape = (N, self.Kf1)
            free = self._dyz_pool.get((N, self.Kf1)) if can_pool else None
            if free:
                dyz = pooled = free.pop()
            elif can_pool:
                dyz = pooled = torch.zeros(N, self.Kf1, dtype=BF16, device=dev)
            else:
                dyz = torch.empty(N, self.Kf1, dtype=BF16, device=dev)
                if self.Kf1 > H + A1:
                    dyz[:, H + A1:].zero_()  # finite values under the zero weight columns
            dt1, dy1 = self._ln_bwd(p + ".attention.output.LayerNorm", da, sv.ln1, run.p_hid, sv.seed_ln1,
                                    dysum=self.G[ad["a1"]["name"] + ".up.bias"], dy_out=dyz[:, :H])
            self._adapter_bwd(run, ad["a1"], dy1, sv.z1, sv.ob, sv.seed_ad1, dz_out=dyz[:, H:H + A1], pooled=pooled)
            L.gemm(dyz, W["WoF"], out_bf16=dctx)
        else:
            dt1, dy1 = self._ln_bwd(p + ".attention.output.LayerNorm", da, sv.ln1, run.p_hid, sv.seed_ln1,
                                    dysum=self.G[ad["a1"]["name"] + ".up.bias"] if "a1" in ad else None)
            do = dy1
            if "a1" in ad:
                do = self._adapter_bwd(run, ad["a1"], dy1, sv.z1, sv.ob, sv.seed_ad1)
            L.gemm(do, W["WoT"], out_bf16=dctx)
        dqkv = self._attn_bwd(run, sv, dctx)
        if not sv.emd:
            dx = torch.empty(N, H, dtype=F32, device=dev)
            L.gemm(dqkv, W["WqkvT"], aux=dt1, aux_kind=L.AUX_ADD_F32, out_f32=dx)
            return dx, None
        # Enhanced mask decoder (model/deberta.py:1382-1412): the query stream and the key/value stream differ.  Returns the
        # gradient of the query stream and, in place of the key/value-stream gradient, the bf16 [dK | dV] operand it is the
        # product of -- the caller chains both passes' products into ONE accumulator through the GEMM epilogues
        # (dx = dq_first + dkv_second + dkv_first: no stand-alone additions of [N, H] fp32 tensors).
        dq = torch.empty(N, H, dtype=F32, device=dev)
        L.gemm(dqkv[:, :H], W["WqkvT"][:, :H], aux=dt1, aux_kind=L.AUX_ADD_F32, out_f32=dq)
        return dq, dqkv[:, H:]

    def _attn_bwd(self, run, sv, dctx):
        B, S, H, nh = run.B, run.S, self.H, self.nh
        N = run.N
        dqkv = torch.empty(N, 3 * H, dtype=BF16, device=self.dev)
        from .attn_bwd import disent_attn_bwd

        if self.reducer is not None:  # ~0.15 ms (0.35 until round 6) without one-workgroup-per-CU GEMM tiles: where gradient collectives may start
            self.reducer.window()

        # The position-table products of this execution (dPK = G1^T.Q, dPQ = G2^T.K) are NOT formed here: the shear passes and
        # the preparation kernel write their operands into this execution's slice of per-step tensors, and one strided-batch
        # chain at the END of backward handles all executions (attn_bwd.pos_table_grads_batched).  ft_ln=False: nothing
        # trainable sits behind the position tables, the operands are scratch.
        pc = getattr(run, "pos_chain", None)
        bufs = None
        if pc is not None:
            e = pc["n"]
            if "G1T" in pc:
                bufs = (pc["G1T"][e], pc["G2T"][e], pc["QT"][e], pc["KT"][e])
            pc["seeds"].append(sv.seed_pos)
            pc["n"] = e + 1
        st = disent_attn_bwd(self, run, sv, dctx, dqkv, None, defer_pos=True, bufs=bufs)
        if pc is not None and "X1" in pc:  # the fused position-gradient kernel reads these at the end of backward
            pc["X1"].append(st["dS"]); pc["X2"].append(st["dST"]); pc["Yq"].append(st["q"]); pc["Yk"].append(st["k"])
            pc["klen"], pc["row0"] = st["klen"], st["row0"]
        return dqkv

    def _head_bwd(self, run, rows, dlog, dq, all_rows=False):
        """Backward of the prediction head for the rows `rows` (int32 indices into the N token rows) given their bf16
        logit gradients dlog [R, Vp]: dq[rows] += d/d(head input).  Head LayerNorm gradients are accumulated."""
        H, dev = self.H, self.dev
        R, Vp = dlog.shape
        Vout = run.Vout
        if Vout == self.V:
            tableT = self.ETb
        else:
            tableT = torch.zeros(H, Vp, dtype=BF16, device=dev)
            tableT[:, :Vout] = self.Ansb.t()
        dhl = L.zeros(R, H, dtype=F32, device=dev)
        if R >= 2048:  # enough rows to fill the chip without splitting K
            L.gemm(dlog, tableT, out_f32=dhl)
        else:  # few rows, long K -> split-K so the grid covers the chip (accumulates into zeros)
            L.gemm(dlog, tableT, out_f32=dhl, splitk=max(2, min(16, Vp // 8192)), ws=self.sk_ws)
        rl = rows.long()
        hn = run.head_norm
        sub = NormRef(hn.t[rl].contiguous(), hn.stats[rl].contiguous(), hn.gamma, hn.beta)
        dt, _ = self._ln_bwd("lm_predictions.lm_head.LayerNorm", dhl, sub, 0.0, 0, want_dy_bf16=False)
        dpre = torch.empty(R, H, dtype=BF16, device=dev)
        L.dropout_gelu_bwd(dt, run.head_pre[rl].contiguous(), 0.0, 0, out_bf16=dpre)
        dqr = torch.empty(R, H, dtype=F32, device=dev)
        L.gemm(dpre, self.WhT, out_f32=dqr)
        if all_rows:
            dq.add_(dqr)
        else:
            L.scatter_rows_f32(dqr, rows, dq)  # first contribution: dq is still zero at these rows

    def backward(self, run, gloss: Optional[torch.Tensor], glogits: Optional[torch.Tensor] = None, attach: bool = True):
        """Explicit backward of _forward; accumulates into the flat gradient buffer (p.grad views).  gloss: gradient
        of the internal MLM loss (or None); glogits: gradient w.r.t. the returned logits [B,S,Vout] (or None).
        attach=False: the caller has already made p.grad the views of the flat buffer (a captured backward must not contain
        the conditional zero fill of attach_grads)."""
        if not run.save:
            raise RuntimeError("forward was run without gradient bookkeeping")
        with L.seed_word(getattr(run, "seed_word", None)):
            return self._backward(run, gloss, glogits, attach)

    def _backward(self, run, gloss, glogits, attach):
        cfg, H, dev = self.cfg, self.H, self.dev
        B, S, T = run.B, run.S, run.T
        N = run.N
        pk = getattr(run, "pk", None)
        if attach:
            self.attach_grads()
        reducer = self.reducer

        red = _Ready(self, run, reducer) if reducer is not None else None
        run.dw_pending, run.dw_ready_keys, run.dw_count = [], [], 0
        dq = L.zeros(N, H, dtype=F32, device=dev)
        Vout = run.Vout
        Vp = _ru(Vout, 64)
        # ---- CE + head, on the labelled rows only (all other rows have exactly zero gradient)
        if gloss is not None and run.labels is not None:
            rows = run.rows_i32
            R = rows.numel()
            if R > 0:
                dlog = torch.empty(R, Vp, dtype=BF16, device=dev)
                # the incoming loss gradient stays on the device (read by the kernel): no host sync at backward start
                gs = gloss.detach().to(F32) if (isinstance(gloss, torch.Tensor) and gloss.is_cuda) else float(gloss)
                ar = torch.arange(R, dtype=torch.int32, device=dev)  # logits_c holds the labelled rows only
                L.ce_bwd_rows(run.logits_c, run.labels_c, ar, Vout, Vp, run.row_lse, run.loss_acc, gs, dlog)
                self._head_bwd(run, rows, dlog, dq)
                del dlog
        # ---- gradient handed in on the logits themselves (downstream losses): every token row
        if glogits is not None:
            gl = glogits.reshape(B * S, Vout)
            if pk is not None:  # (gradients handed in at positions without a row have nothing to flow into)
                gl = gl.index_select(0, pk.sel)
            dlog = torch.zeros(N, Vp, dtype=BF16, device=dev) if Vp != Vout else torch.empty(N, Vp, dtype=BF16, device=dev)
            dlog[:, :Vout].copy_(gl)
            self._head_bwd(run, torch.arange(N, dtype=torch.int32, device=dev), dlog, dq, all_rows=True)
            del dlog

        def stage_done(key):  # a stage's gradients are final once the adapter products parked in it have been launched
            run.dw_ready_keys.append(key)
            self._dw_flush(run, red)

        stage_done("head")
        run.pos_chain = None
        if "deberta.encoder.LayerNorm.weight" in self.G:
            from .attn_bwd import pos_chain_buffers

            run.pos_chain = pos_chain_buffers(self, run, len(run.layers))
        # ---- EMD: two executions of the last layer, newest first
        layers = run.layers
        dkv_ops = []
        for _ in range(2):
            sv = layers.pop()
            dq, dkv_op = self._layer_bwd(run, sv, dq)
            dkv_ops.append(dkv_op)
        # q0 = pos_emb + hs[-2]: the query-stream gradient flows into hs[-2] too, next to both passes' key/value-stream
        # gradients [dK | dV] . [Wk ; Wv] -- accumulated by the epilogues of those two GEMMs
        WkvT = self.Lw[self.nL - 1]["WqkvT"][:, H:]
        dx = dq
        for op in dkv_ops:
            acc = torch.empty(N, H, dtype=F32, device=dev)
            L.gemm(op, WkvT, aux=dx, aux_kind=L.AUX_ADD_F32, out_f32=acc)
            dx = acc
        del dkv_ops
        stage_done(f"layer{self.nL - 1}")
        # ---- encoder layers nL-2 .. 0
        while layers:
            sv = layers.pop()
            if sv.li == 0 and cfg.conv_kernel_size:
                dx, dcol = self._conv_bwd(run, dx)
                dx, _ = self._layer_bwd(run, sv, dx)
                if pk is None:
                    L.col2im3(dcol, dx, B, S, H, 1)
                else:  # packed rows: fold the three taps on the padded grid, add the rows that exist
                    dcg = torch.zeros(B * S, 3 * H, dtype=F32, device=dev).index_copy_(0, pk.sel, dcol)
                    dxg = torch.zeros(B * S, H, dtype=F32, device=dev)
                    L.col2im3(dcg, dxg, B, S, H, 1)
                    dx.add_(dxg.index_select(0, pk.sel))
                    del dcg, dxg
                stage_done("conv")
            else:
                dx, _ = self._layer_bwd(run, sv, dx)
            stage_done(f"layer{sv.li}")
        self._dw_flush(run, red, force=True)
        if getattr(run, "side_used", False):
            torch.cuda.current_stream().wait_stream(self.side)  # all adapter dW/db are in the flat grad buffer
        # ---- relative-position LayerNorm (receives grads from every layer execution)
        rn = run.rel_norm
        if run.pos_chain is not None:
            from .attn_bwd import pos_table_grads_batched

            dR = pos_table_grads_batched(self, run, run.pos_chain)
            run.pos_chain = None
            L.ln_bwd(dR, rn.t, rn.stats, rn.gamma, dgamma=self.G.get("deberta.encoder.LayerNorm.weight"),
                     dbeta=self.G.get("deberta.encoder.LayerNorm.bias"), ws=self._ln_ws)
        if red:
            red.ready("relln")
        # ---- embeddings: dropout -> *mask -> LN ; linear_video
        if run.p_hid > 0:
            L.dropout_f32(dx, run.p_hid, run.seed_emb, out_f32=dx)
        en = run.emb_norm
        dt0 = torch.empty(N, H, dtype=F32, device=dev)
        L.ln_bwd(dx, en.t, en.stats, en.gamma, rowmask=en.rowmask, out_dt=dt0,
                 dgamma=self.G.get("deberta.embeddings.LayerNorm.weight"), dbeta=self.G.get("deberta.embeddings.LayerNorm.bias"),
                 ws=self._ln_ws)
        if T:
            if pk is None:
                dv = dt0.view(B, S, H)[:, :T].reshape(B * T, H).contiguous()
            else:  # the video slots are the first T rows of every sample
                dv = dt0.index_select(0, (pk.row0[:-1].long()[:, None] + torch.arange(T, device=dev)[None]).reshape(-1))
            Kp = _ru(B * T, 64)
            dvT = torch.empty(H, Kp, dtype=BF16, device=dev)
            vT = torch.empty(self.Fp, Kp, dtype=BF16, device=dev)
            L.transpose_to_bf16(dv, dvT)
            L.transpose_to_bf16(run.video_bf16, vT)
            L.gemm(dvT, vT, aux=self.G["deberta.embeddings.linear_video.weight"], aux_kind=L.AUX_ADD_F32,
                   out_f32=self.G["deberta.embeddings.linear_video.weight"], N=self.F)
            L.colsum(dv, self.G["deberta.embeddings.linear_video.bias"], self._cs_ws)
        if red:
            red.ready("emb")
            red.finish()

    def _conv_bwd(self, run, dout):
        """Backward of _conv_fwd: returns (grad of the layer-0 output, grad of the im2col matrix)."""
        B, S, H, dev = run.B, run.S, self.H, self.dev
        N = run.N
        cn = run.conv_norm
        dt = torch.empty(N, H, dtype=F32, device=dev)
        L.ln_bwd(dout, cn.t, cn.stats, cn.gamma, rowmask=cn.rowmask, out_dt=dt,
                 dgamma=self.G.get("deberta.encoder.conv.LayerNorm.weight"),
                 dbeta=self.G.get("deberta.encoder.conv.LayerNorm.bias"), ws=self._ln_ws)
        dc = torch.empty(N, H, dtype=BF16, device=dev)
        L.dropout_gelu_bwd(dt, run.conv_c, run.p_hid, run.seed_conv, out_bf16=dc)
        dcol = torch.empty(N, 3 * H, dtype=F32, device=dev)
        L.gemm(dc, self.WcT, rowscale=run.mask_f, out_f32=dcol)
        return dt, dcol


class _Ready:
    """Reducer front end used by Engine.backward: a bucket may only leave once the side-stream dW kernels that fill it
    have finished.  (Module level on purpose: a class created per call is only reclaimed by the cyclic GC, and its
    closure would keep the whole `run` -- gigabytes of saved activations -- alive until then.)"""

    __slots__ = ("eng", "run", "reducer")

    def __init__(self, eng, run, reducer):
        self.eng, self.run, self.reducer = eng, run, reducer

    def ready(self, key):
        ev = getattr(self.run, "dw_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        self.reducer.ready(key)

    def finish(self):
        self.reducer.finish()


@dataclass
class LayerSave:
    li: int = 0
    emd: bool = False
    qkv: torch.Tensor = None
    pqk: torch.Tensor = None
    ctx: torch.Tensor = None
    lse: torch.Tensor = None
    ob: torch.Tensor = None
    z1: torch.Tensor = None
    ln1: NormRef = None
    hpre: torch.Tensor = None
    fb: torch.Tensor = None
    z2: torch.Tensor = None
    ln2: NormRef = None
    seed_pos: int = 0
    seed_att: int = 0
    seed_ad1: int = 0
    seed_ad2: int = 0
    seed_ln1: int = 0
    seed_ln2: int = 0
    psave: torch.Tensor = None
    msave: torch.Tensor = None


@dataclass
class Run:
    B: int
    S: int
    T: int
    Lt: int
    train: bool
    save: bool
    seed_base: int
    p_hid: float
    p_att: float
    p_ad: float
    layers: List[LayerSave] = field(default_factory=list)
    _site: int = 0
    mask: torch.Tensor = None
    labels: torch.Tensor = None
    hidden_out: tuple = None
    seed_word: Optional[torch.Tensor] = None  # device word added to every dropout seed of this pass (see Engine.run)
    want_attn: bool = False
    attn_out: list = field(default_factory=list)
    seed_emb: int = 0
    seed_conv: int = 0
    pk: Optional["Packing"] = None  # packed-row layout of this pass (model.packed_rows) or None: the padded [B, S] grid
    N: int = 0                      # activation rows: B*S, or pk.n

    def __post_init__(self):
        if not self.N:
            self.N = self.B * self.S

    def next_seed(self) -> int:
        self._site += 1
        # seed_base: the step's position in the mask stream (eager) or 0 (captured: the device word carries it; Run.seed_word)
        return (self.seed_base + self._site * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF


class _StepFn(torch.autograd.Function):
    """Single autograd node for the whole model: forward already ran; backward runs Engine.backward which writes the
    gradients straight into the flat grad buffer (p.grad views), so autograd itself accumulates nothing."""

    @staticmethod
    def forward(ctx, engine, run, loss_t, logits_t, *params):
        ctx.engine, ctx.run = engine, run
        ctx.set_materialize_grads(False)
        return loss_t.detach().clone(), logits_t.detach()

    @staticmethod
    def backward(ctx, gloss, glogits):
        eng, run = ctx.engine, ctx.run
        if gloss is None and glogits is None:
            return (None, None, None, None) + tuple(None for _ in eng.order)
        eng.backward(run, gloss, glogits)
        return (None, None, None, None) + tuple(None for _ in eng.order)
