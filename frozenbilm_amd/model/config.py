"""Literal DeBERTa-v2 configuration (no hub access on the GPU box).  Field names follow transformers'
DebertaV2Config as read by the reference (model/deberta.py:445-472, 683-696, 960-981); defaults are the
microsoft/deberta-v2-xlarge values."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List


@dataclass
class DebertaV2Config:
    vocab_size: int = 128100
    hidden_size: int = 1536
    num_hidden_layers: int = 24
    num_attention_heads: int = 24
    intermediate_size: int = 6144
    max_position_embeddings: int = 512
    type_vocab_size: int = 0
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    layer_norm_eps: float = 1e-7
    initializer_range: float = 0.02
    relative_attention: bool = True
    position_buckets: int = 256
    max_relative_positions: int = -1
    pos_att_type: List[str] = field(default_factory=lambda: ["p2c", "c2p"])
    norm_rel_ebd: str = "layer_norm"
    share_att_key: bool = True
    position_biased_input: bool = False
    conv_kernel_size: int = 3
    conv_act: str = "gelu"
    pad_token_id: int = 0
    use_return_dict: bool = True

    @classmethod
    def from_any(cls, cfg) -> "DebertaV2Config":
        """Accept our own config, a transformers config object or a dict."""
        if isinstance(cfg, cls):
            return cfg
        src = cfg if isinstance(cfg, dict) else {k: getattr(cfg, k) for k in cls.__dataclass_fields__ if hasattr(cfg, k)}
        out = cls(**{k: v for k, v in src.items() if k in cls.__dataclass_fields__ and v is not None})
        if isinstance(out.pos_att_type, str):
            out.pos_att_type = [x.strip() for x in out.pos_att_type.lower().split("|")]
        return out

    def validate_supported(self):
        """The MI355X path implements exactly the DeBERTa-v2 variant FrozenBiLM uses."""
        hd = self.hidden_size // self.num_attention_heads
        problems = []
        if hd != 64 or self.hidden_size % self.num_attention_heads:
            problems.append(f"head_dim must be 64 (got {hd})")
        if not self.relative_attention or sorted(self.pos_att_type) != ["c2p", "p2c"] or not self.share_att_key:
            problems.append("needs relative_attention with pos_att_type {c2p,p2c} and share_att_key")
        if "layer_norm" not in self.norm_rel_ebd:
            problems.append("needs norm_rel_ebd=layer_norm")
        if self.position_biased_input or self.type_vocab_size:
            problems.append("position_biased_input / token types are not part of the path")
        if self.conv_kernel_size not in (0, 3):
            problems.append("conv_kernel_size must be 0 or 3")
        if self.hidden_act != "gelu" or (self.conv_kernel_size and self.conv_act != "gelu"):
            problems.append("activations must be erf-gelu")
        if self.hidden_size % 64 or self.intermediate_size % 64:
            problems.append("hidden/intermediate sizes must be multiples of 64")
        if problems:
            raise NotImplementedError("unsupported DeBERTa-v2 configuration: " + "; ".join(problems))

    @property
    def att_span(self) -> int:
        if self.position_buckets > 0:
            return self.position_buckets
        return self.max_relative_positions if self.max_relative_positions >= 1 else self.max_position_embeddings

    @property
    def max_rel(self) -> int:
        return self.max_relative_positions if self.max_relative_positions >= 1 else self.max_position_embeddings
