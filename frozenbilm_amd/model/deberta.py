"""MI355X-native DebertaV2ForMaskedLM with video prefix + adapters: drop-in for the reference's
model/deberta.py:1289-1501 on the masked-LM path.

Same constructor and ``forward`` signature, same ``state_dict`` key names, same freeze policy
(model/deberta.py:1152-1158, 1334-1339).  Underneath there is no ATen math: ``forward`` runs the explicit HIP pipeline
of ``frozenbilm_amd.engine`` (C-ABI kernels of libfbl.so on the current HIP stream) and, when gradients are enabled,
returns a loss whose ``backward()`` runs the explicit backward pipeline (one autograd node for the whole model).

Host-side layout (288 GB HBM: replicate freely):
  * fp32 masters under the reference parameter names; every TRAINABLE parameter (linear_video, adapters, LayerNorms)
    is a view into ONE flat fp32 buffer, its gradient a view into one flat grad buffer ordered by backward completion
    (head LN, layer 23 ... layer 0, conv LN, encoder LN, embeddings) -> fused clip+Adam and bucketed RCCL all-reduce
    work on contiguous slices.
  * frozen matrices are packed once to bf16 MFMA operands, both W ([out,in], forward) and W^T (dX = dY.W).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import lib as L
from .adapter import Adapter  # noqa: F401  (re-export, mirrors `from model.adapter import Adapter`)
from .config import DebertaV2Config


class MaskedLMOutput(dict):
    """Attribute + item access (`out.loss`, `out["loss"]`), like transformers' MaskedLMOutput (main.py:67).

    When a loss was asked for (``labels`` given) the forward computes only the labelled rows of the prediction head;
    the full ``logits`` tensor is allocated -- and wired into the autograd graph like the reference's -- but its
    contents are produced on first access, by any accessor of this mapping (``out.logits``, ``out["logits"]``,
    ``out[1]``, ``get``, ``values``, ``items``, iteration / ``dict(out)`` / ``**out``)."""

    _fill = None  # callable that materialises the logits, or None

    def _sync_logits(self):
        fill = self.__dict__.get("_fill")
        if fill is not None:
            self.__dict__["_fill"] = None
            fill()

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            self._sync_logits()
            return [v for v in super().values() if v is not None][k]
        if k == "logits":
            self._sync_logits()
        return super().__getitem__(k)

    def get(self, k, default=None):
        if k == "logits":
            self._sync_logits()
        return super().get(k, default)

    def __iter__(self):  # (also takes dict(out) / {**out} off CPython's fast path, which would bypass __getitem__)
        self._sync_logits()
        return super().__iter__()

    def keys(self):
        self._sync_logits()
        return super().keys()

    def values(self):
        self._sync_logits()
        return super().values()

    def items(self):
        self._sync_logits()
        return super().items()

    def copy(self):
        self._sync_logits()
        return dict(super().items())

    def pop(self, k, *default):
        if k == "logits":
            self._sync_logits()
        return super().pop(k, *default)


class _Node(nn.Module):
    """Bare container used to reproduce the reference's module tree (hence its state_dict key names)."""


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def param_shapes(cfg: DebertaV2Config, features_dim: int, ds_attn: int, ds_ff: int, n_ans: int) -> "OrderedDict[str, tuple]":
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    e = "deberta.embeddings"
    sh[e + ".word_embeddings.weight"] = (V, H)
    sh[e + ".position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    sh[e + ".LayerNorm.weight"] = (H,)
    sh[e + ".LayerNorm.bias"] = (H,)
    if features_dim:
        sh[e + ".linear_video.weight"] = (H, features_dim)
        sh[e + ".linear_video.bias"] = (H,)
    for i in range(cfg.num_hidden_layers):
        p = f"deberta.encoder.layer.{i}"
        for n in ("query_proj", "key_proj", "value_proj"):
            sh[f"{p}.attention.self.{n}.weight"] = (H, H)
            sh[f"{p}.attention.self.{n}.bias"] = (H,)
        sh[p + ".attention.output.dense.weight"] = (H, H)
        sh[p + ".attention.output.dense.bias"] = (H,)
        sh[p + ".attention.output.LayerNorm.weight"] = (H,)
        sh[p + ".attention.output.LayerNorm.bias"] = (H,)
        if ds_attn:
            A = H // ds_attn
            sh[p + ".attention.output.adapter.down.weight"] = (A, H)
            sh[p + ".attention.output.adapter.down.bias"] = (A,)
            sh[p + ".attention.output.adapter.up.weight"] = (H, A)
            sh[p + ".attention.output.adapter.up.bias"] = (H,)
        sh[p + ".intermediate.dense.weight"] = (I, H)
        sh[p + ".intermediate.dense.bias"] = (I,)
        sh[p + ".output.dense.weight"] = (H, I)
        sh[p + ".output.dense.bias"] = (H,)
        sh[p + ".output.LayerNorm.weight"] = (H,)
        sh[p + ".output.LayerNorm.bias"] = (H,)
        if ds_ff:
            A = H // ds_ff
            sh[p + ".output.adapter.down.weight"] = (A, H)
            sh[p + ".output.adapter.down.bias"] = (A,)
            sh[p + ".output.adapter.up.weight"] = (H, A)
            sh[p + ".output.adapter.up.bias"] = (H,)
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
    if n_ans:
        sh["answer_embeddings.weight"] = (n_ans, H)
        sh["answer_bias"] = (n_ans,)
    return sh


def flat_order(cfg: DebertaV2Config, names: List[str]) -> List[str]:
    """Trainable names in backward-completion order (bucket order of the gradient all-reduce)."""
    def take(prefix):
        return [n for n in names if n.startswith(prefix)]

    out: List[str] = []
    out += take("lm_predictions.")
    for i in reversed(range(cfg.num_hidden_layers)):
        if i == 0:
            out += take("deberta.encoder.conv.")
        out += take(f"deberta.encoder.layer.{i}.")
    out += take("deberta.encoder.LayerNorm.")
    out += take("deberta.embeddings.")
    rest = [n for n in names if n not in set(out)]
    return out + rest


class DebertaV2ForMaskedLM(nn.Module):
    def __init__(
        self,
        config,
        max_feats=10,
        features_dim=768,
        freeze_lm=True,
        freeze_mlm=True,
        ds_factor_attn=8,
        ds_factor_ff=8,
        ft_ln=True,
        dropout=0.1,
        n_ans=0,
        freeze_last=True,
    ):
        super().__init__()
        self.config = DebertaV2Config.from_any(config)
        self.config.validate_supported()
        cfg = self.config
        for ds in (ds_factor_attn, ds_factor_ff):
            if ds:
                assert not cfg.hidden_size % ds  # model/adapter.py:10
        self.max_feats = max_feats
        self.features_dim = features_dim
        self.ds_factor_attn = ds_factor_attn
        self.ds_factor_ff = ds_factor_ff
        self.adapter_dropout = float(dropout) if dropout else 0.0
        self.n_ans = n_ans
        self.freeze_lm, self.freeze_mlm, self.ft_ln, self.freeze_last = freeze_lm, freeze_mlm, ft_ln, freeze_last
        if not (freeze_lm and freeze_mlm):
            raise NotImplementedError(
                "the MI355X path implements FrozenBiLM's frozen-LM regime (freeze_lm=freeze_mlm=True): "
                "only linear_video, adapters and LayerNorms receive gradients")
        if n_ans and not freeze_last:
            raise NotImplementedError("--ft_last (trainable answer-embedding module) is not implemented: the answer "
                                      "head is used frozen, the reference's default (args.py:358-363)")

        shapes = param_shapes(cfg, features_dim, ds_factor_attn, ds_factor_ff, n_ans)
        g = torch.Generator().manual_seed(0)
        for name, shape in shapes.items():
            if "LayerNorm" in name:
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif name.endswith(".bias") or name == "answer_bias":
                t = torch.zeros(shape)  # model/deberta.py:1085-1086
            else:
                t = torch.randn(shape, generator=g) * cfg.initializer_range  # :1084,1088
            if name.endswith("word_embeddings.weight"):
                t[cfg.pad_token_id].zero_()  # :1089-1090
            self._register(name, nn.Parameter(t, requires_grad=self._trainable(name)))
        emb = self._module("deberta.embeddings")
        emb.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)))
        self._engine = None
        self.engine_options = {}  # read once when the engine is built (engine.Engine.__init__ lists the keys; defaults = shipped)
        self.inference_graphs = False  # opt-in: replay the `logit_rows` inference forward as one hipGraph (_graph_forward)
        self.training_graphs = False  # opt-in: the MLM training step as two replayed hipGraphs (train_graph.py)
        # opt-in: ragged batches run without the padding rows behind each sample's last used position (engine.Packing) whenever
        # the call's outputs live on selected rows (labels / logit_rows given).  Rows that exist get the same values as on the
        # padded grid; `logits` / `hidden_states` read as zero at the dropped positions (the reference computes values there
        # that nothing on this path reads), training-mode dropout draws a different -- equally distributed -- mask stream, and
        # the launch graphs (shape-static) are not used while it is on.
        self.packed_rows = False
        self._weights_frozen = 0  # nesting depth of weights_frozen(): inference forwards may reuse the packed operands
        self._reducer = None  # parallel.GradReducer attached to this model (survives engine rebuilds)
        self.step_seed = 0  # advanced every training forward; keys the counter-based dropout
        self._seed_salt = None  # torch seed (args.seed + rank in the reference's main()) and rank, fixed at first use

    # ---------------------------------------------------------------- module tree helpers
    def _module(self, dotted: str) -> nn.Module:
        m: nn.Module = self
        for part in dotted.split("."):
            if part not in m._modules:
                m.add_module(part, _Node())
            m = m._modules[part]
        return m

    def _register(self, name: str, p: nn.Parameter):
        if "." in name:
            mod, leaf = name.rsplit(".", 1)
            self._module(mod).register_parameter(leaf, p)
        else:
            self.register_parameter(name, p)

    def _trainable(self, name: str) -> bool:
        if name.startswith("answer_"):
            return not self.freeze_last
        if "linear_video" in name or "adapter" in name:
            return True
        return bool(self.ft_ln and "LayerNorm" in name)

    # ---------------------------------------------------------------- reference API
    @property
    def device(self):
        return next(self.parameters()).device

    def get_param(self, name: str) -> torch.Tensor:
        m: nn.Module = self
        parts = name.split(".")
        for part in parts[:-1]:
            m = m._modules[part]
        return m._parameters[parts[-1]]

    def set_answer_embeddings(self, a2tok, freeze_last=True):
        """model/deberta.py:1358-1380: answer table = masked mean of the word embeddings of each answer's tokens.
        (The reference assigns ``answer_bias.weight``, an attribute, so the effective bias keeps its value.)"""
        if not freeze_last:
            raise NotImplementedError("--ft_last (trainable answer-embedding module) is not implemented")
        E = self.get_param("deberta.embeddings.word_embeddings.weight")
        pad = self.config.pad_token_id
        a2tok = a2tok.to(E.device)
        keep = (a2tok != pad)
        table = (E.data[a2tok] * keep.float()[:, :, None]).sum(1) / keep.sum(1, keepdim=True).clamp(min=1)
        if len(table) != self.n_ans or "answer_embeddings" not in self._modules:
            assert not self.training
            self.n_ans = len(table)
            self._register("answer_embeddings.weight", nn.Parameter(table.clone(), requires_grad=False))
            old = self._parameters.get("answer_bias")
            self.register_parameter("answer_bias", nn.Parameter(torch.zeros(self.n_ans, device=E.device), requires_grad=False))
            del old
        else:
            self.get_param("answer_embeddings.weight").data = table
        self.freeze_last = freeze_last
        self.get_param("answer_embeddings.weight").requires_grad_(False)
        self.get_param("answer_bias").requires_grad_(False)
        self.invalidate()

    def weights_frozen(self):
        """Context manager: inside it the caller promises not to modify the trainable parameters behind the engine's back
        (through `.data`, `vector_to_parameters`, raw writes into `engine().flat`), so inference forwards skip the rebuild
        of the bf16 operands / composed adapter rows unless FusedAdam.step or an in-place update through a parameter object
        happened in between.  Outside it every forward rebuilds them.  The evaluate loops run inside it."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self._weights_frozen += 1
            try:
                yield self
            finally:
                self._weights_frozen -= 1
                if self._weights_frozen == 0 and self._engine is not None:
                    self._engine._ops_version = None

        return scope()

    def invalidate(self):
        """Drop the packed bf16 operands / flat buffers (after load_state_dict, .to(), set_answer_embeddings)."""
        self._engine = None
        self.__dict__.pop("_graph_cache", None)  # captured graphs hold the old engine's buffers
        self.__dict__.pop("_train_graphs", None)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self.invalidate()
        # tolerate the reference-only keys (untied decoder copy, buffers)
        sd = {k: v for k, v in state_dict.items() if "lm_head.decoder" not in k}
        return super().load_state_dict(sd, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self._engine is None:
            from ..engine import Engine

            self._engine = Engine(self)
            if self._reducer is not None:
                self._reducer.rebind(self._engine)
        return self._engine

    def dropout_seed_base(self) -> int:
        """Seed of this training step's dropout sites.  The reference draws its masks from the per-process torch RNG,
        seeded with args.seed + rank (main.py:161-165): mixing torch.initial_seed() and the rank gives every data-parallel
        rank and every differently seeded run its own mask stream; ``step_seed`` (saved in checkpoints) advances it."""
        if self._seed_salt is None:
            rank = 0
            try:
                import torch.distributed as dist

                if dist.is_available() and dist.is_initialized():
                    rank = dist.get_rank()
            except Exception:
                rank = 0
            self._seed_salt = ((torch.initial_seed() & 0xFFFFFFFF) * 2654435761 + rank * 40503 + 12345) & 0xFFFFFFFFFFFF
        return (self.step_seed * 1000003 + self._seed_salt) & 0xFFFFFFFFFFFF

    # ---------------------------------------------------------------- inference launch graphs (opt-in)
    GRAPH_L_BUCKET = 32  # text lengths are padded up to a multiple of this inside the graph path
    GRAPH_CACHE = 6      # captured (batch, length, rows) configurations kept per model

    def _graph_forward(self, eng, input_ids, attention_mask, video, video_mask, mlm, logit_rows):
        """The `logit_rows` inference forward (what `videoqa.evaluate` / `mc.evaluate` run per batch, videoqa.py:157-168,
        mc.py:160-172) as ONE hipGraph replay: ~700 kernel launches cost the host ~10 ms per batch when issued one by one,
        more than the loops have to spare next to an 18 ms forward.  The text is padded to the next multiple of
        GRAPH_L_BUCKET with pad tokens (masked: a sample's logits do not depend on padding), the inputs are copied into
        the graph's static buffers, the graph -- captured on first use of a (batch, padded length, frames, rows)
        configuration, least recently used ones dropped -- is replayed and the [rows, V] logits are returned as a copy.
        Returns None when the configuration cannot be captured (the caller then takes the eager path)."""
        import torch.nn.functional as F_

        # operands of the trainable weights / composed adapter rows are rebuilt (in place) outside the graph: on every call,
        # or -- inside weights_frozen(), where the evaluate loops run -- only when something wrote to the parameters
        eng.prepare_inference()
        B, Lt = input_ids.shape
        T = video.shape[1] if (video is not None and eng.F) else 0
        Lp = min(-(-Lt // self.GRAPH_L_BUCKET) * self.GRAPH_L_BUCKET, self.config.max_position_embeddings - T)
        if Lp < Lt:
            return None
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        rows = logit_rows.to(eng.dev).long().view(-1)
        S_old, S_new = T + Lt, T + Lp
        rows = (rows // S_old) * S_new + rows % S_old
        key = (id(eng), B, Lp, T, video_mask is not None, rows.numel(), bool(mlm))
        cache = self.__dict__.setdefault("_graph_cache", {})
        feed = dict(input_ids=F_.pad(input_ids, (0, Lp - Lt), value=self.config.pad_token_id),
                    attention_mask=F_.pad(attention_mask, (0, Lp - Lt), value=0), rows=rows.to(torch.int32))
        if T:
            feed["video"] = video
            if video_mask is not None:
                feed["video_mask"] = video_mask
        ent = cache.pop(key, None)
        if ent is None:
            for k in [k for k in cache if k[0] != id(eng)]:
                del cache[k]
            while len(cache) >= self.GRAPH_CACHE:
                del cache[next(iter(cache))]
            static = {k: v.to(eng.dev).clone() for k, v in feed.items()}

            def run_once():
                return eng.run(static["input_ids"], static["attention_mask"], static.get("video"), static.get("video_mask"),
                               None, mlm, False, logit_rows=static["rows"])["logits"]

            eng._no_pos_cache = True  # the captured forward recomputes the position projections: it reads nothing that a later
            try:                      # rebuild of the trainable operands would re-allocate behind its back
                run_once()  # warm-up outside the capture (lazy initialisation, allocator)
                torch.cuda.synchronize(eng.dev)
                graph = torch.cuda.CUDAGraph()
                import gc

                gc_was_on = gc.isenabled()  # (no cyclic collection inside a capture: train_graph.TrainStep.capture)
                gc.collect()
                gc.disable()
                try:
                    with torch.cuda.graph(graph):
                        logits = run_once()
                finally:
                    if gc_was_on:
                        gc.enable()
            except Exception as e:  # noqa: BLE001
                import warnings

                torch.cuda.synchronize(eng.dev)
                warnings.warn(f"inference graph capture failed ({type(e).__name__}: {e}); staying on the eager path")
                self.inference_graphs = False
                return None
            finally:
                eng._no_pos_cache = False
            ent = (graph, static, logits)
        graph, static, logits = ent
        cache[key] = ent  # most recently used last
        for k, v in feed.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return logits.clone()

    def forward(
        self,
        input_ids=None,
        attention_mask=None,
        token_type_ids=None,
        position_ids=None,
        inputs_embeds=None,
        labels=None,
        output_attentions=None,
        return_dict=None,
        video=None,
        video_mask=None,
        mlm=False,
        output_hidden_states=False,
        logit_rows=None,
    ):
        """Reference signature (model/deberta.py:1414-1427) plus two keyword extensions: ``output_hidden_states`` and
        ``logit_rows`` -- int tensor of flat row indices b*S + s into the [B, S] token grid (S = video slots + text): at
        inference the prediction head then runs on those rows only and ``logits`` is [len(logit_rows), V] (the downstream
        loops read one [MASK] row per sample: videoqa.py:164-168, mc.py:166-170)."""
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None:
            if inputs_embeds is not None:
                raise NotImplementedError("inputs_embeds is not on the FrozenBiLM hot path")
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        eng = self.engine()
        if (self.inference_graphs and not self.packed_rows and logit_rows is not None and labels is None and not output_hidden_states
                and not output_attentions and not self.training and not torch.is_grad_enabled()):
            logits = self._graph_forward(eng, input_ids, attention_mask, video, video_mask, mlm, logit_rows)
            if logits is not None:
                out = MaskedLMOutput(loss=None, logits=logits, hidden_states=None, attentions=None)
                return out if return_dict is not False else (logits,)
        if (not self.training_graphs and eng.reducer is not None and "_overlap_before_graphs" in eng.reducer.__dict__
                and not self.__dict__.get("_train_graphs_auto_off")):
            from ..train_graph import restore_reducer_overlap

            restore_reducer_overlap(eng)  # graphs were on earlier in this run: give the reducer its overlap placement back
        if (self.training_graphs and not self.packed_rows and self.training and labels is not None and not output_hidden_states
                and not output_attentions
                and logit_rows is None and not (self.n_ans and not mlm) and torch.is_grad_enabled()):
            from ..train_graph import graphed_forward

            got = graphed_forward(self, eng, input_ids, attention_mask, video, video_mask, labels)
            if got is not None:
                loss, run = got
                B, S = run.B, run.S
                # (the logits of a graphed step carry no autograd edge: gradients flow through `.loss`)
                out = MaskedLMOutput(loss=loss, logits=run.logits.view(B, S, -1)[:, :, : run.Vout], hidden_states=None,
                                     attentions=None)
                out.__dict__["_run"] = run
                out.__dict__["_fill"] = lambda: eng.fill_logits(run)
                return out if return_dict is not False else (out["loss"], out["logits"])
        # output_attentions=True (model/deberta.py:1414-1427, :544-560): the fused attention kernel never materialises its
        # probabilities; on request a plain kernel rebuilds them per encoder layer from the stored log-sum-exp (eval mode)
        res = eng.run(input_ids, attention_mask, video, video_mask, labels, mlm, output_hidden_states, logit_rows=logit_rows,
                      want_attn=bool(output_attentions))
        out = MaskedLMOutput(loss=res["loss"], logits=res["logits"], hidden_states=res.get("hidden_states"),
                             attentions=res.get("attentions"))
        run = res.get("run")
        out.__dict__["_run"] = run  # (tests read the saved bottleneck activations through this)
        if run is not None and getattr(run, "logits_pending", False):
            out.__dict__["_fill"] = lambda: eng.fill_logits(run)
        if return_dict is False:
            return tuple(v for v in (out["loss"], out["logits"], out["hidden_states"], out["attentions"]) if v is not None)
        return out
