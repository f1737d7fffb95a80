"""CPU restatement (torch fp32) of the reference's BERT variant -- BASELINE config 1, SURVEY.md row a25.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): vanilla BERT encoder + the ``linear_video`` projection that prepends
CLIP frame features to the token stream, tied MLM decoder.  The reference's BERT has no adapters (only the freeze rule
mentions them, ``model/bert.py:549``).  Every function cites the reference lines it follows; pinned by golden G8
(tests/golden/make_goldens.py imports ``model/bert.py`` with the shims of SURVEY App. B item 6).

Parameter names are the reference's ``state_dict`` keys.  ``cls.predictions.decoder.weight`` is tied to
``bert.embeddings.word_embeddings.weight`` (transformers 4.17 ``tie_weights``), so it is not a separate entry here.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclasses.dataclass
class BertOracleConfig:
    """``BertConfig()`` defaults = BERT-base (config 1: vocab 30522, no adapters, features_dim 768)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0
    features_dim: int = 768
    max_feats: int = 10
    n_ans: int = 0


def param_shapes(cfg: BertOracleConfig) -> Dict[str, tuple]:
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s: Dict[str, tuple] = {}
    e = "bert.embeddings."
    s[e + "word_embeddings.weight"] = (V, H)
    s[e + "position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    if cfg.features_dim:
        s[e + "linear_video.weight"] = (H, cfg.features_dim)
        s[e + "linear_video.bias"] = (H,)
    for i in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    c = "cls.predictions."
    s[c + "bias"] = (V,)
    s[c + "transform.dense.weight"] = (H, H)
    s[c + "transform.dense.bias"] = (H,)
    s[c + "transform.LayerNorm.weight"] = (H,)
    s[c + "transform.LayerNorm.bias"] = (H,)
    if cfg.n_ans:
        s["answer_embeddings.weight"] = (cfg.n_ans, H)
        s["answer_bias"] = (cfg.n_ans,)
    return s


def synth_params(cfg: BertOracleConfig, seed: int = 0, std: float = 0.02, ln_jitter: float = 0.0) -> Params:
    """Same recipe as ``deberta_oracle.synth_params``: N(0, std) tensor by tensor in ``param_shapes`` order."""
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


def _ln(x, P, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], eps)


def embeddings(cfg, P, input_ids, video=None):
    """``BertEmbeddings.forward`` (model/bert.py:242-278): cat(linear_video(video), word_emb) + absolute position
    embeddings (positions 0..S-1 over the CONCATENATED sequence) + token-type-0 embeddings, LayerNorm (dropout: eval)."""
    e = "bert.embeddings."
    x = P[e + "word_embeddings.weight"][input_ids]
    if cfg.features_dim and video is not None:
        v = F.linear(video, P[e + "linear_video.weight"], P[e + "linear_video.bias"])  # :238-240
        x = torch.cat([v, x], 1)
    S = x.shape[1]
    x = x + P[e + "position_embeddings.weight"][:S][None] + P[e + "token_type_embeddings.weight"][0][None, None]
    return _ln(x, P, e + "LayerNorm", cfg.layer_norm_eps)


def self_attention(cfg, P, prefix, x, ext_mask):
    """``BertSelfAttention.forward`` (model/bert.py:138-191): softmax(QK^T/sqrt(d) + additive mask) V."""
    B, S, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh

    def heads(t):
        return t.view(B, S, nh, d).permute(0, 2, 1, 3)

    q = heads(F.linear(x, P[prefix + "query.weight"], P[prefix + "query.bias"]))
    k = heads(F.linear(x, P[prefix + "key.weight"], P[prefix + "key.bias"]))
    v = heads(F.linear(x, P[prefix + "value.weight"], P[prefix + "value.bias"]))
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + ext_mask
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, S, H)


def layer(cfg, P, i, x, ext_mask):
    """``BertLayer.forward`` (model/bert.py:379-428) = BertAttention (:325-346, BertSelfOutput :288-292) ->
    BertIntermediate (:203-206, erf-GELU) -> BertOutput (:356-360); post-LN residual blocks."""
    p = f"bert.encoder.layer.{i}."
    ctx = self_attention(cfg, P, p + "attention.self.", x, ext_mask)
    a = F.linear(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"])
    a = _ln(a + x, P, p + "attention.output.LayerNorm", cfg.layer_norm_eps)
    h = F.gelu(F.linear(a, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"]))
    o = F.linear(h, P[p + "output.dense.weight"], P[p + "output.dense.bias"])
    return _ln(o + a, P, p + "output.LayerNorm", cfg.layer_norm_eps)


def extended_mask(attention_mask):
    """``get_extended_attention_mask`` as pinned by the reference (transformers 4.17): (1 - mask) * -10000 broadcast
    over heads and query rows (model/bert.py:640-642).  Unlike DeBERTa's XSoftmax nothing zeroes masked QUERY rows."""
    return (1.0 - attention_mask[:, None, None, :].float()) * -10000.0


def forward(cfg: BertOracleConfig, P: Params, input_ids, attention_mask=None, video=None, video_mask=None,
            labels=None, mlm: bool = False):
    """``BertForMaskedLM.forward`` (model/bert.py:792-872) / ``BertModel.forward`` (:571-700), eval mode.
    Returns dict(loss, logits[B,S,V or n_ans], hidden)."""
    B, L = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones(B, L)
    use_video = bool(cfg.features_dim) and video is not None
    if use_video:
        if video_mask is None:
            video_mask = torch.ones(video.shape[:2])
        attention_mask = torch.cat([video_mask.to(attention_mask.dtype), attention_mask], 1)  # :629-634
    ext = extended_mask(attention_mask)
    x = embeddings(cfg, P, input_ids, video if use_video else None)
    for i in range(cfg.num_hidden_layers):
        x = layer(cfg, P, i, x, ext)
    c = "cls.predictions."
    t = F.gelu(F.linear(x, P[c + "transform.dense.weight"], P[c + "transform.dense.bias"]))  # :67-71
    t = _ln(t, P, c + "transform.LayerNorm", cfg.layer_norm_eps)
    if cfg.n_ans and not mlm:  # downstream mode (:88-95, :837-840)
        logits = t @ P["answer_embeddings.weight"].t() + P["answer_bias"]
    else:  # tied decoder + output-only bias (:74-86)
        logits = F.linear(t, P["bert.embeddings.word_embeddings.weight"], P[c + "bias"])
    loss: Optional[torch.Tensor] = None
    if labels is not None:
        if use_video:  # :844-853: visual slots are never predicted
            labels = torch.cat([torch.full(video.shape[:2], -100, dtype=torch.long), labels], 1)
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
    return dict(loss=loss, logits=logits, hidden=x)
