"""CPU oracle (torch fp32 + numpy) of the FrozenBiLM DeBERTa-v2 MLM path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Functional restatement:
parameters live in a flat ``dict[str, Tensor]`` keyed by the reference's
``state_dict`` names, every function cites the reference
lines it follows (paths relative to /root/reference).

All floating point math is fp32 on CPU, integer tables are int64 built through
float64 exactly as the reference does.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class OracleConfig:
    """DeBERTa-v2 hyper-parameters read by model/deberta.py."""

    vocab_size: int = 128100
    hidden_size: int = 1536
    num_hidden_layers: int = 24
    num_attention_heads: int = 24
    intermediate_size: int = 6144
    max_position_embeddings: int = 512
    position_buckets: int = 256
    max_relative_positions: int = -1
    layer_norm_eps: float = 1e-7
    conv_kernel_size: int = 3
    pad_token_id: int = 0
    # FrozenBiLM additions (model/deberta.py:1293-1306)
    features_dim: int = 1024
    max_feats: int = 10
    ds_factor_attn: int = 8
    ds_factor_ff: int = 8
    n_ans: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def att_span(self) -> int:  # model/deberta.py:694-696,841
        return self.position_buckets if self.position_buckets > 0 else self.max_rel

    @property
    def max_rel(self) -> int:  # model/deberta.py:448-450
        return (
            self.max_relative_positions
            if self.max_relative_positions >= 1
            else self.max_position_embeddings
        )


# --------------------------------------------------------------------------
# integer tables
# --------------------------------------------------------------------------
def log_bucket(rel: np.ndarray, bucket_size: int, max_position: int) -> np.ndarray:
    """model/deberta.py:578-589 make_log_bucket_position (float64 log, int64 out)."""
    rel = np.asarray(rel, dtype=np.int64)
    mid = bucket_size // 2
    sgn = np.sign(rel)
    inside = (rel < mid) & (rel > -mid)
    a = np.where(inside, mid - 1, np.abs(rel))
    lp = np.ceil(np.log(a / mid) / np.log((max_position - 1) / mid) * (mid - 1)) + mid
    return np.where(a <= mid, rel, lp * sgn).astype(np.int64)


def relative_position(q: int, k: int, bucket_size: int, max_position: int) -> np.ndarray:
    """model/deberta.py:592-618 build_relative_position -> int64 [q, k] (delta = i - j)."""
    delta = np.arange(q, dtype=np.int64)[:, None] - np.arange(k, dtype=np.int64)[None, :]
    if bucket_size > 0 and max_position > 0:
        delta = log_bucket(delta, bucket_size, max_position)
    return delta


def rel_index_by_delta(S: int, cfg: OracleConfig) -> np.ndarray:
    """Toeplitz index vector: out[d + S - 1] = clamp(bucket(d) + span, 0, 2*span-1), d in [-(S-1), S-1].

    This is the c2p index of model/deberta.py:873 seen as a function of delta=i-j.
    """
    d = np.arange(-(S - 1), S, dtype=np.int64)
    b = log_bucket(d, cfg.position_buckets, cfg.max_rel) if cfg.position_buckets > 0 else d
    span = cfg.att_span
    return np.clip(b + span, 0, 2 * span - 1).astype(np.int64)


# --------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------
# Optional "bf16 operand" mode (test infrastructure): every matrix-multiply operand is rounded to bfloat16 on the way
# in (straight-through in backward), accumulation stays fp32 -- the arithmetic contract of the HIP path (bf16 MFMA
# operands, fp32 accumulators, fp32 residual stream / LayerNorm / softmax).  With the same rounding the ReLU gates of
# the adapters and the GELU arguments agree with the GPU's except at exact ties, so gradient parity can be held to a
# few per cent instead of the 25 % a gate flip costs against the pure fp32 math.  Default off: the fp32 mode is what
# the golden vectors captured from the reference pin.
_BF16_OPERANDS = False


class bf16_operands:
    def __enter__(self):
        global _BF16_OPERANDS
        self._prev, _BF16_OPERANDS = _BF16_OPERANDS, True
        return self

    def __exit__(self, *exc):
        global _BF16_OPERANDS
        _BF16_OPERANDS = self._prev
        return False


def _rb(x: torch.Tensor) -> torch.Tensor:
    if not _BF16_OPERANDS:
        return x
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def _lin(x: torch.Tensor, P: Params, name: str) -> torch.Tensor:
    return F.linear(_rb(x), _rb(P[name + ".weight"]), P.get(name + ".bias"))


def _ln(x: torch.Tensor, P: Params, name: str, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """transformers ACT2FN["gelu"]: exact erf GELU (model/deberta.py:305-308)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# Optional ReLU-gate override (test infrastructure): inside ``with adapter_gates(masks)`` every adapter call consumes the
# next 0/1 mask of the list (execution order: per layer execution the attention-output adapter, then the FFN-output one)
# and uses it in place of relu'(.) -- forward and backward.  A test takes the masks from the bottleneck activations the
# GPU run saved (z > 0), which removes the one discontinuity through which rounding noise becomes an O(10 %) difference
# in d(down.weight): what is left must agree to rounding.
_GATES = None


class adapter_gates:
    def __init__(self, masks):
        self.masks = list(masks)

    def __enter__(self):
        global _GATES
        self._prev, _GATES = _GATES, iter(self.masks)
        return self

    def __exit__(self, *exc):
        global _GATES
        _GATES = self._prev
        return False


# Optional dropout-mask injection (test infrastructure): the reference trains with StableDropout / nn.Dropout live at six
# sites per layer execution plus the embeddings, the convolution branch, the attention probabilities and the shared
# relative-position table (model/deberta.py:142-240, 258, 332, 403, 779, 796, 1054; model/adapter.py:40-41).  Inside
# ``with dropout_masks(provider)`` every one of those sites asks ``provider(kind, shape)`` -- in the REFERENCE's execution
# order -- for a multiplicative mask (0 = dropped, 1/(1-p) = kept; None = no dropout at this site) and applies it, forward
# and backward (XDropout.backward :185-190 multiplies the gradient by the same mask).  kinds: "emb" (:1054), "pos" (:779),
# "att" (:796), "ad" (adapter bottleneck, adapter.py:41), "hid" (SelfOutput :258 / Output :332), "conv" (:403).
# Two users: the golden G17 (masks drawn from a seeded generator exactly as the patched reference drew them: pins this
# train-mode restatement) and the GPU train-mode parity tests (masks rebuilt from the HIP path's counter-based RNG).
_DROP = None


class dropout_masks:
    def __init__(self, provider):
        self.provider = provider

    def __enter__(self):
        global _DROP
        self._prev, _DROP = _DROP, self.provider
        return self

    def __exit__(self, *exc):
        global _DROP
        _DROP = self._prev
        return False


def _drop(x: torch.Tensor, kind: str) -> torch.Tensor:
    if _DROP is None:
        return x
    m = _DROP(kind, tuple(x.shape))
    return x if m is None else x * m.to(x.dtype).view(x.shape)


def adapter(x: torch.Tensor, P: Params, prefix: str, drop: Optional[torch.Tensor] = None) -> torch.Tensor:
    """model/adapter.py:33-45 with default flags: x + up(drop(relu(down(x)))).

    ``drop`` (optional) is a multiplicative keep/scale mask [.., A] standing in for nn.Dropout.
    """
    pre = _lin(x, P, prefix + ".down")
    if _GATES is not None:
        z = pre * next(_GATES).to(pre.dtype).view(pre.shape)
    else:
        z = torch.relu(pre)
    if drop is not None:
        z = z * drop
    z = _drop(z, "ad")  # model/adapter.py:40-41
    return x + _lin(z, P, prefix + ".up")


def _split_heads(x: torch.Tensor, nh: int) -> torch.Tensor:
    """model/deberta.py:712-715 transpose_for_scores: [B,S,H] -> [B*nh, S, d]."""
    B, S, H = x.shape
    return x.view(B, S, nh, H // nh).permute(0, 2, 1, 3).reshape(B * nh, S, H // nh)


def disentangled_attention(
    hidden: torch.Tensor,
    mask4d: torch.Tensor,
    rel_pos: np.ndarray,
    rel_emb: torch.Tensor,
    P: Params,
    prefix: str,
    cfg: OracleConfig,
    query_states: Optional[torch.Tensor] = None,
    return_probs: bool = False,
):
    """model/deberta.py:717-818 (forward) + :820-947 (bias), pos_att_type = {c2p, p2c}, share_att_key.

    hidden [B,S,H]; mask4d [B,1,S,S] (0/1); rel_pos int64 [S,S] bucketed i-j;
    rel_emb [2*span, H] = LayerNorm_enc(rel_embeddings.weight).
    """
    nh, d = cfg.num_attention_heads, cfg.head_dim
    B, S, H = hidden.shape
    q_in = hidden if query_states is None else query_states
    q = _rb(_split_heads(_lin(q_in, P, prefix + ".query_proj"), nh))  # :757-759
    k = _rb(_split_heads(_lin(hidden, P, prefix + ".key_proj"), nh))  # :760-762
    v = _rb(_split_heads(_lin(hidden, P, prefix + ".value_proj"), nh))  # :763-765
    scale = math.sqrt(d * 3)  # :769-776 scale_factor = 1 + c2p + p2c
    scores = torch.bmm(q, k.transpose(1, 2)) / scale  # :777

    span = cfg.att_span
    rel = torch.from_numpy(rel_pos).long()
    rel_emb = _drop(rel_emb, "pos")  # :779 pos_dropout on the shared table, a fresh mask per layer execution
    pos_q = _rb(_split_heads(_lin(rel_emb[None], P, prefix + ".query_proj"), nh))  # :848-850 [nh,2span,d]
    pos_k = _rb(_split_heads(_lin(rel_emb[None], P, prefix + ".key_proj"), nh))  # :851-853
    pos_q = pos_q.repeat(B, 1, 1)
    pos_k = pos_k.repeat(B, 1, 1)
    # c2p :870-881
    c2p = torch.bmm(q, pos_k.transpose(1, 2))  # [BH,S,2span]
    c2p_idx = torch.clamp(rel + span, 0, 2 * span - 1)  # [S,S]
    c2p = torch.gather(c2p, 2, c2p_idx[None].expand(B * nh, S, S))
    bias = c2p / scale
    # p2c :884-918
    p2c_idx = torch.clamp(-rel + span, 0, 2 * span - 1)
    p2c = torch.bmm(k, pos_q.transpose(1, 2))  # [BH,S(j),2span]
    p2c = torch.gather(p2c, 2, p2c_idx[None].expand(B * nh, S, S)).transpose(1, 2)
    bias = bias + p2c / scale
    scores = (scores + bias).view(B, nh, S, S)

    # XSoftmax :123-132
    rmask = ~(mask4d.bool())
    probs = torch.softmax(scores.masked_fill(rmask, float("-inf")), -1)
    probs = probs.masked_fill(rmask, 0.0)
    probs_out = probs
    probs = _drop(probs, "att")  # :796
    ctx = torch.bmm(_rb(probs.view(B * nh, S, S)), v)  # :797-802
    ctx = ctx.view(B, nh, S, d).permute(0, 2, 1, 3).reshape(B, S, H)  # :803-814
    if return_probs:
        return ctx, probs_out
    return ctx


def layer(
    hidden: torch.Tensor,
    mask4d: torch.Tensor,
    rel_pos: np.ndarray,
    rel_emb: torch.Tensor,
    P: Params,
    prefix: str,
    cfg: OracleConfig,
    query_states: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """model/deberta.py:351-375 DebertaV2Layer (attention :271-297, SelfOutput :254-260,
    Intermediate :310-313, Output :328-334); dropout sites are identities unless masks are injected (dropout_masks)."""
    eps = cfg.layer_norm_eps
    ctx = disentangled_attention(hidden, mask4d, rel_pos, rel_emb, P, prefix + ".attention.self", cfg, query_states)
    resid = hidden if query_states is None else query_states  # :290-292
    o = _lin(ctx, P, prefix + ".attention.output.dense")
    if cfg.ds_factor_attn:
        o = adapter(o, P, prefix + ".attention.output.adapter")
    o = _drop(o, "hid")  # :258
    a = _ln(o + resid, P, prefix + ".attention.output.LayerNorm", eps)
    h = gelu_erf(_lin(a, P, prefix + ".intermediate.dense"))
    f = _lin(h, P, prefix + ".output.dense")
    if cfg.ds_factor_ff:
        f = adapter(f, P, prefix + ".output.adapter")
    f = _drop(f, "hid")  # :332
    return _ln(f + a, P, prefix + ".output.LayerNorm", eps)


def conv_layer(emb: torch.Tensor, resid: torch.Tensor, input_mask: torch.Tensor, P: Params, cfg: OracleConfig) -> torch.Tensor:
    """model/deberta.py:395-419 ConvLayer (conv_act = gelu, groups = 1), eval mode."""
    pre = "deberta.encoder.conv"
    pad = (cfg.conv_kernel_size - 1) // 2
    c = F.conv1d(_rb(emb.permute(0, 2, 1).contiguous()), _rb(P[pre + ".conv.weight"]), P[pre + ".conv.bias"], padding=pad)
    c = c.permute(0, 2, 1).contiguous()
    c = c.masked_fill((1 - input_mask).bool()[..., None], 0.0)
    c = _drop(c, "conv")  # :403
    out = _ln(resid + gelu_erf(c), P, pre + ".LayerNorm", cfg.layer_norm_eps)
    return out * input_mask[..., None].to(out.dtype)


def embeddings(input_ids, video, mask, P: Params, cfg: OracleConfig):
    """model/deberta.py:997-1058 (position_biased_input False, type_vocab 0): returns (emb, pos_emb[1,S,H])."""
    pre = "deberta.embeddings"
    x = F.embedding(input_ids, P[pre + ".word_embeddings.weight"])
    if cfg.features_dim and video is not None:
        x = torch.cat([_lin(video, P, pre + ".linear_video"), x], 1)  # :1013-1015
    S = x.shape[1]
    pos = P[pre + ".position_embeddings.weight"][:S][None]  # :1020-1029 (position_ids = arange)
    x = _ln(x, P, pre + ".LayerNorm", cfg.layer_norm_eps)
    x = x * mask[..., None].to(x.dtype)  # :1045-1052
    x = _drop(x, "emb")  # :1054
    return x, pos


def encoder(emb: torch.Tensor, mask: torch.Tensor, P: Params, cfg: OracleConfig, skip_dead_last: bool = False):
    """model/deberta.py:507-575: returns list of hidden states (25 entries for 24 layers)."""
    B, S, H = emb.shape
    m = mask.to(torch.float32)
    mask4d = (m[:, None, None, :] * m[:, None, :, None]).to(torch.uint8)  # :480-490
    rel_pos = relative_position(S, S, cfg.position_buckets, cfg.max_rel)  # :492-505
    rel_emb = _ln(P["deberta.encoder.rel_embeddings.weight"], P, "deberta.encoder.LayerNorm", cfg.layer_norm_eps)  # :474-478
    hs = [emb]
    x = emb
    L = cfg.num_hidden_layers
    for i in range(L):
        if skip_dead_last and i == L - 1:
            hs.append(None)
            break
        x = layer(x, mask4d, rel_pos, rel_emb, P, f"deberta.encoder.layer.{i}", cfg)
        if i == 0 and cfg.conv_kernel_size > 0:
            x = conv_layer(emb, x, mask, P, cfg)  # :549-550 (input = embeddings, residual = layer-0 output)
        hs.append(x)
    return hs, mask4d, rel_pos, rel_emb


def emd(hs: List[torch.Tensor], pos_emb: torch.Tensor, mask4d, rel_pos, rel_emb, P: Params, cfg: OracleConfig) -> torch.Tensor:
    """model/deberta.py:1382-1412: two passes of the last layer, query = pos_emb + hs[-2]."""
    hidden = hs[-2]
    q = pos_emb.expand_as(hidden) + hidden  # :1392 (z_states += hidden_states)
    last = f"deberta.encoder.layer.{cfg.num_hidden_layers - 1}"
    for _ in range(2):
        q = layer(hidden, mask4d, rel_pos, rel_emb, P, last, cfg, query_states=q)
    return q


def lm_head(x: torch.Tensor, table: torch.Tensor, bias: torch.Tensor, P: Params, cfg: OracleConfig) -> torch.Tensor:
    """model/deberta.py:1544-1558: LN(gelu(dense(x))) @ table^T + bias."""
    pre = "lm_predictions.lm_head"
    h = gelu_erf(_lin(x, P, pre + ".dense"))
    h = _ln(h, P, pre + ".LayerNorm", cfg.layer_norm_eps)
    return _rb(h) @ _rb(table).t() + bias


def answer_embeddings(a2tok: torch.Tensor, P: Params, cfg: OracleConfig) -> torch.Tensor:
    """model/deberta.py:1358-1373 set_answer_embeddings: masked mean of word embeddings of answer tokens."""
    E = P["deberta.embeddings.word_embeddings.weight"]
    a2v = F.embedding(a2tok, E)
    keep = (a2tok != cfg.pad_token_id)
    n = keep.sum(1, keepdim=True).clamp(min=1)
    return (a2v * keep.float()[:, :, None]).sum(1) / n


def forward(
    P: Params,
    cfg: OracleConfig,
    input_ids: torch.Tensor,
    attention_mask: torch.Tensor,
    video: Optional[torch.Tensor] = None,
    video_mask: Optional[torch.Tensor] = None,
    labels: Optional[torch.Tensor] = None,
    mlm: bool = False,
    return_hidden: bool = False,
):
    """model/deberta.py:1414-1501 DebertaV2ForMaskedLM.forward (eval mode) -> dict(loss, logits[, hidden_states])."""
    B = input_ids.shape[0]
    if cfg.features_dim and video is not None:
        if video_mask is None:
            video_mask = torch.ones(video.shape[:2], dtype=attention_mask.dtype)
        mask = torch.cat([video_mask.to(attention_mask.dtype), attention_mask], 1)  # :1220-1225
    else:
        mask = attention_mask
    emb, pos = embeddings(input_ids, video, mask, P, cfg)
    hs, mask4d, rel_pos, rel_emb = encoder(emb, mask, P, cfg, skip_dead_last=not return_hidden)
    out = emd(hs, pos, mask4d, rel_pos, rel_emb, P, cfg)
    if cfg.n_ans and not mlm:  # :1474-1478
        table = P["answer_embeddings.weight"]
        bias = P["answer_bias"]
    else:
        table = P["deberta.embeddings.word_embeddings.weight"]
        bias = P["lm_predictions.lm_head.bias"]
    logits = lm_head(out, table, bias, P, cfg)
    loss = None
    if labels is not None:
        if cfg.features_dim and video is not None:  # :1452-1462
            labels = torch.cat([torch.full(video.shape[:2], -100, dtype=torch.long), labels], 1)
        loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100)
    res = {"loss": loss, "logits": logits}
    if return_hidden:
        res["hidden_states"] = hs
    return res


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of the reference model, buffers excluded."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    sh: Dict[str, Tuple[int, ...]] = {}
    e = "deberta.embeddings"
    sh[e + ".word_embeddings.weight"] = (V, H)
    sh[e + ".position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    sh[e + ".LayerNorm.weight"] = (H,)
    sh[e + ".LayerNorm.bias"] = (H,)
    if cfg.features_dim:
        sh[e + ".linear_video.weight"] = (H, cfg.features_dim)
        sh[e + ".linear_video.bias"] = (H,)
    for i in range(cfg.num_hidden_layers):
        p = f"deberta.encoder.layer.{i}"
        for n in ("query_proj", "key_proj", "value_proj"):
            sh[f"{p}.attention.self.{n}.weight"] = (H, H)
            sh[f"{p}.attention.self.{n}.bias"] = (H,)
        for blk, din, ds in ((".attention.output", H, cfg.ds_factor_attn), (".output", I, cfg.ds_factor_ff)):
            sh[p + blk + ".dense.weight"] = (H, din)
            sh[p + blk + ".dense.bias"] = (H,)
            sh[p + blk + ".LayerNorm.weight"] = (H,)
            sh[p + blk + ".LayerNorm.bias"] = (H,)
            if ds:
                A = H // ds
                sh[p + blk + ".adapter.down.weight"] = (A, H)
                sh[p + blk + ".adapter.down.bias"] = (A,)
                sh[p + blk + ".adapter.up.weight"] = (H, A)
                sh[p + blk + ".adapter.up.bias"] = (H,)
        sh[p + ".intermediate.dense.weight"] = (I, H)
        sh[p + ".intermediate.dense.bias"] = (I,)
    c = "deberta.encoder"
    sh[c + ".rel_embeddings.weight"] = (2 * cfg.att_span, H)
    sh[c + ".LayerNorm.weight"] = (H,)
    sh[c + ".LayerNorm.bias"] = (H,)
    if cfg.conv_kernel_size > 0:
        sh[c + ".conv.conv.weight"] = (H, H, cfg.conv_kernel_size)
        sh[c + ".conv.conv.bias"] = (H,)
        sh[c + ".conv.LayerNorm.weight"] = (H,)
        sh[c + ".conv.LayerNorm.bias"] = (H,)
    h = "lm_predictions.lm_head"
    sh[h + ".bias"] = (V,)
    sh[h + ".dense.weight"] = (H, H)
    sh[h + ".dense.bias"] = (H,)
    sh[h + ".LayerNorm.weight"] = (H,)
    sh[h + ".LayerNorm.bias"] = (H,)
    if cfg.n_ans:
        sh["answer_embeddings.weight"] = (cfg.n_ans, H)
        sh["answer_bias"] = (cfg.n_ans,)
    return sh


def is_trainable(name: str, ft_ln: bool = True) -> bool:
    """Freeze policy of model/deberta.py:1152-1158 + :1334-1339 (freeze_lm, freeze_mlm, ft_ln defaults)."""
    if "linear_video" in name or "adapter" in name:
        return True
    if ft_ln and "LayerNorm" in name:
        return True
    return False


def synth_params(cfg: OracleConfig, seed: int = 0, std: float = 0.02, ln_jitter: float = 0.0) -> Params:
    """Seeded synthetic parameters (SURVEY.md section 8d recipe): N(0, std) matrices/embeddings/biases
    generated tensor-by-tensor in ``param_shapes`` order from one CPU generator; LayerNorm weight 1,
    bias 0 (plus optional N(0, ln_jitter) so tests exercise gamma/beta)."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in param_shapes(cfg).items():
        if "LayerNorm" in name:
            base = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            if ln_jitter:
                base = base + ln_jitter * torch.randn(shape, generator=g)
            P[name] = base
        else:
            P[name] = torch.randn(shape, generator=g) * std
    return P
