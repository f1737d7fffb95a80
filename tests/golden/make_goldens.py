"""Generate the golden vectors under tests/golden/ by importing the REFERENCE (read-only, /root/reference).

Run only in the build container (the reference does not exist on the GPU box):

    python tests/golden/make_goldens.py [--only G5] [--skip-xl]

The reference ships no tests, so these vectors -- inputs plus the reference's own outputs -- are what pins
the oracle (oracle/) and, through it, the HIP path.  Only DATA is written (npz of inputs/weights/outputs);
no reference source is copied.  Import shims follow SURVEY.md App. B.
"""
from __future__ import annotations

import argparse
import dataclasses
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))


def load_reference():
    import transformers.modeling_utils as mu

    mu.PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)  # shim 1
    np.int = int  # shim 2
    pkg = types.ModuleType("model")
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules["model"] = pkg

    def _load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    adapter = _load("model.adapter", os.path.join(REF, "model/adapter.py"))
    deberta = _load("model.deberta", os.path.join(REF, "model/deberta.py"))
    deberta.BaseModelOutput = dataclasses.dataclass(deberta.BaseModelOutput)  # shim 4
    deberta._softmax_backward_data = lambda g, o, dim, t: torch._softmax_backward_data(g, o, dim, t.dtype)  # shim 5
    misc = _load("ref_util_misc", os.path.join(REF, "util/misc.py"))
    return adapter, deberta, misc


def ref_config(cfg):
    from transformers import DebertaV2Config

    return DebertaV2Config(
        vocab_size=cfg.vocab_size,
        hidden_size=cfg.hidden_size,
        num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads,
        intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings,
        type_vocab_size=0,
        hidden_act="gelu",
        hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1,
        layer_norm_eps=cfg.layer_norm_eps,
        relative_attention=True,
        position_buckets=cfg.position_buckets,
        max_relative_positions=cfg.max_relative_positions,
        pos_att_type=["p2c", "c2p"],
        norm_rel_ebd="layer_norm",
        share_att_key=True,
        position_biased_input=False,
        conv_kernel_size=cfg.conv_kernel_size,
        conv_act="gelu",
        pad_token_id=cfg.pad_token_id,
        attention_head_size=cfg.hidden_size // cfg.num_attention_heads,
    )


def build_ref_model(deberta, cfg, P, ft_ln=True):
    m = deberta.DebertaV2ForMaskedLM(
        ref_config(cfg),
        max_feats=cfg.max_feats,
        features_dim=cfg.features_dim,
        ds_factor_attn=cfg.ds_factor_attn,
        ds_factor_ff=cfg.ds_factor_ff,
        n_ans=cfg.n_ans,
        ft_ln=ft_ln,
    )
    sd = m.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    extra = [k for k in sd if k not in P and "position_ids" not in k and "decoder" not in k]
    assert not extra, extra
    m.load_state_dict({k: v.clone() for k, v in P.items()}, strict=False)
    m.eval()
    return m


def synth_batch(cfg, B, L, seed, ragged=True, lo_id=5):
    """Seeded synthetic batch: fp16-rounded N(0,1) features, ragged lengths, 15% labels."""
    g = torch.Generator().manual_seed(seed)
    T = cfg.max_feats
    video = torch.randn(B, T, cfg.features_dim, generator=g).half().float()
    if ragged:
        vlen = torch.randint(1, T + 1, (B,), generator=g)
        vlen[0] = T
        tlen = torch.randint(max(2, L // 8), L + 1, (B,), generator=g)
        tlen[-1] = L
    else:
        vlen = torch.full((B,), T)
        tlen = torch.full((B,), L)
    ids = torch.randint(lo_id, cfg.vocab_size, (B, L), generator=g)
    pos = torch.arange(L)[None]
    amask = (pos < tlen[:, None]).long()
    ids = ids * amask  # pad id 0
    vmask = (torch.arange(T)[None] < vlen[:, None]).long()
    sel = (torch.rand(B, L, generator=g) < 0.15) & amask.bool()
    sel[:, 1] = True  # at least one label per row
    labels = torch.where(sel, ids, torch.full_like(ids, -100))
    return dict(video=video, video_mask=vmask, input_ids=ids, attention_mask=amask, labels=labels)


def npz(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KB)")


# ----------------------------------------------------------------------------------------------
def g1_adapter(adapter_mod):
    torch.manual_seed(11)
    a = adapter_mod.Adapter(8, 128, dropout=0.1).eval()
    with torch.no_grad():
        for p in a.parameters():
            p.copy_(torch.randn(p.shape) * 0.05)
    x = torch.randn(2, 7, 128)
    y = a(x)
    npz("G1_adapter", x=x, y=y, **{"w." + k: v for k, v in a.state_dict().items()})


def g2_relpos(deberta):
    out = {}
    for S in (1, 2, 74, 129, 266, 512):
        r = deberta.build_relative_position(S, S, bucket_size=256, max_position=512)[0].numpy()
        assert (r == r[:, :1] * 0 + r).all()
        # Toeplitz: store first column (delta >= 0) and first row (delta <= 0) + a checksum of the full table
        out[f"col_{S}"] = r[:, 0].copy()
        out[f"row_{S}"] = r[0, :].copy()
        out[f"sum_{S}"] = np.array([int(np.abs(r).sum()), int((r * np.arange(S)[None, :]).sum())], dtype=np.int64)
        i, j = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
        d = i - j
        tv = np.where(d >= 0, r[:, 0][np.abs(d)], r[0, :][np.abs(d)])
        assert (tv == r).all(), "reference table is not Toeplitz"
    npz("G2_relpos", **out)


def _tiny_cfg(**kw):
    from oracle.deberta_oracle import OracleConfig

    base = dict(vocab_size=512, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=512,
                features_dim=32, max_feats=10)
    base.update(kw)
    return OracleConfig(**base)


def g3_attention(deberta):
    """DisentangledSelfAttention eval, d=64, 2 heads, S=37 and 266, padded rows, with and without query_states."""
    from oracle.deberta_oracle import synth_params, relative_position

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    enc = m.deberta.encoder
    att = enc.layer[1].attention.self
    out = {}
    for S in (37, 266):
        g = torch.Generator().manual_seed(100 + S)
        B = 2
        hidden = torch.randn(B, S, cfg.hidden_size, generator=g)
        qs = torch.randn(B, S, cfg.hidden_size, generator=g)
        mask = torch.ones(B, S, dtype=torch.long)
        mask[0, S - 5:] = 0
        mask[1, 3:6] = 0
        with torch.no_grad():
            m4 = enc.get_attention_mask(mask)
            rel = enc.get_rel_pos(hidden)
            remb = enc.get_rel_embedding()
            y0 = att(hidden, m4, False, query_states=None, relative_pos=rel, rel_embeddings=remb)
            y1 = att(hidden, m4, False, query_states=qs, relative_pos=None, rel_embeddings=remb)
        out.update({f"hidden_{S}": hidden, f"qs_{S}": qs, f"mask_{S}": mask, f"ctx_{S}": y0, f"ctxq_{S}": y1})
    out["rel_emb"] = remb
    npz("G3_attention", seed=np.array([3]), **out)


def g3b_attention_backward(deberta):
    """Backward of DisentangledSelfAttention (eval, d=64, 2 heads, S=37 and 266, padded rows, with and without
    query_states; same modules / seeds / inputs as G3): for a seeded upstream gradient dy of the context layer, the
    reference's gradients w.r.t. the OUTPUTS of query_proj / key_proj / value_proj on the token rows (what the HIP backward
    kernels return as dq, dk, dv) and on the relative-position table rows (dPQ, dPK), captured with forward hooks; plus the
    end-to-end input gradients (hidden_states, query_states, rel_embeddings).  S=266: token-row tensors stored every third
    row, fp16 (file size)."""
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    enc = m.deberta.encoder
    att = enc.layer[1].attention.self
    caught = {"q": [], "k": [], "v": []}
    hooks = []
    for key, mod in (("q", att.query_proj), ("k", att.key_proj), ("v", att.value_proj)):
        def hook(_m, _i, o, key=key):
            o.retain_grad()
            caught[key].append(o)
        hooks.append(mod.register_forward_hook(hook))
    out = {}
    for S in (37, 266):
        g = torch.Generator().manual_seed(100 + S)
        B = 2
        hidden = torch.randn(B, S, cfg.hidden_size, generator=g)
        qs = torch.randn(B, S, cfg.hidden_size, generator=g)
        mask = torch.ones(B, S, dtype=torch.long)
        mask[0, S - 5:] = 0
        mask[1, 3:6] = 0
        dy = torch.randn(B, S, cfg.hidden_size, generator=torch.Generator().manual_seed(500 + S))
        for tag, use_q in (("", False), ("q", True)):
            for v in caught.values():
                v.clear()
            h = hidden.clone().requires_grad_(True)
            q_in = qs.clone().requires_grad_(True) if use_q else None
            with torch.no_grad():
                m4 = enc.get_attention_mask(mask)
                rel = enc.get_rel_pos(hidden)
            remb = enc.get_rel_embedding().detach().clone().requires_grad_(True)
            y = att(h, m4, False, query_states=q_in, relative_pos=rel, rel_embeddings=remb)
            y.backward(dy)
            # call order (model/deberta.py:757-765, 847-853): query_proj(tokens), key_proj(tokens), value_proj(tokens),
            # query_proj(rel_embeddings), key_proj(rel_embeddings)
            assert [len(caught[k]) for k in "qkv"] == [2, 2, 1]
            sl = (lambda t: t) if S == 37 else (lambda t: t[:, ::3].half())
            cv = (lambda t: t) if S == 37 else (lambda t: t.half())
            out.update({f"dq{tag}_{S}": sl(caught["q"][0].grad), f"dk{tag}_{S}": sl(caught["k"][0].grad),
                        f"dv{tag}_{S}": sl(caught["v"][0].grad), f"dpq{tag}_{S}": cv(caught["q"][1].grad[0]),
                        f"dpk{tag}_{S}": cv(caught["k"][1].grad[0]), f"dhidden{tag}_{S}": sl(h.grad), f"drel{tag}_{S}": cv(remb.grad)})
            if use_q:
                out[f"dqs_{S}"] = sl(q_in.grad)
        out[f"dy_{S}"] = dy
    for hk in hooks:
        hk.remove()
    npz("G3b_attention_backward", seed=np.array([3]), **out)


def g4_layer_conv(deberta):
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=4, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    enc = m.deberta.encoder
    g = torch.Generator().manual_seed(44)
    B, S = 2, 45
    hidden = torch.randn(B, S, cfg.hidden_size, generator=g)
    qs = torch.randn(B, S, cfg.hidden_size, generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, 30:] = 0
    with torch.no_grad():
        m4 = enc.get_attention_mask(mask)
        rel = enc.get_rel_pos(hidden)
        remb = enc.get_rel_embedding()
        y_layer = enc.layer[2](hidden, m4, False, query_states=None, relative_pos=rel, rel_embeddings=remb)
        y_layer_q = enc.layer[2](hidden, m4, False, query_states=qs, relative_pos=None, rel_embeddings=remb)
        y_conv = enc.conv(hidden, qs, mask)
    npz("G4_layer_conv", seed=np.array([4]), hidden=hidden, qs=qs, mask=mask, y_layer=y_layer, y_layer_q=y_layer_q,
        y_conv=y_conv)


def g5_tiny_model(deberta):
    """Tiny full model eval: logits, loss, all trainable grads, hidden states."""
    from oracle.deberta_oracle import synth_params, is_trainable

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    batch = synth_batch(cfg, B=3, L=27, seed=55)
    out = m(**batch)
    out.loss.backward()
    grads = {}
    names_trainable = []
    for n, p in m.named_parameters():
        if p.requires_grad:
            names_trainable.append(n)
            assert p.grad is not None, n
            grads["grad." + n] = p.grad
    assert sorted(names_trainable) == sorted(k for k in P if is_trainable(k)), "freeze policy mismatch"
    hs = torch.stack(out.hidden_states, 0)
    npz("G5_tiny_model", seed=np.array([5]), logits=out.logits, loss=out.loss, hidden_states=hs,
        **{"in." + k: v for k, v in batch.items()}, **grads)
    # same inputs, no video / no labels (text-only branch)
    with torch.no_grad():
        o2 = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"])
    npz("G5b_tiny_textonly", logits=o2.logits)



class _GenDropout(torch.nn.Module):
    """nn.Dropout stand-in whose keep decisions come from a caller-owned generator (Adapter.dropout, model/adapter.py:21)"""

    def __init__(self, p, gen):
        super().__init__()
        self.p, self.gen = p, gen

    def forward(self, x):
        if not self.training or self.p <= 0:
            return x
        keep = torch.empty_like(x).bernoulli_(1 - self.p, generator=self.gen)
        return x * keep / (1 - self.p)


def seeded_train_mode(deberta, adapter_mod, m, gen):
    """Make every dropout site of the reference model draw from `gen`: StableDropout goes through the module-level
    get_mask (model/deberta.py:151-168), the adapters own an nn.Dropout (model/adapter.py:21).  Returns an undo function."""
    orig = deberta.get_mask

    def get_mask(input, local_context):
        assert not isinstance(local_context, deberta.DropoutContext)  # (context_stack is None on this path)
        dropout = local_context
        mask = None
        if dropout > 0:
            mask = (1 - torch.empty_like(input).bernoulli_(1 - dropout, generator=gen)).bool()
        return mask, dropout

    deberta.get_mask = get_mask
    for mod in m.modules():
        if isinstance(mod, adapter_mod.Adapter) and isinstance(mod.dropout, torch.nn.Dropout):
            mod.dropout = _GenDropout(mod.dropout.p, gen)

    def undo():
        deberta.get_mask = orig

    return undo


def g17_train_mode(deberta, adapter_mod):
    """The reference in train() mode (main.py:34) on the tiny model of G5 with every dropout decision drawn from ONE seeded
    generator in the reference's own execution order: loss, logits, all trainable gradients.  The oracle, handed masks drawn
    from an identically seeded generator in ITS execution order (oracle.dropout_masks), must reproduce them -- which pins the
    place, the scale and the order of all dropout sites of the train-mode restatement (tests/test_oracle_golden.py)."""
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    m.train()
    gen = torch.Generator().manual_seed(1717)
    undo = seeded_train_mode(deberta, adapter_mod, m, gen)
    try:
        batch = synth_batch(cfg, B=3, L=27, seed=55)
        out = m(**batch)
        out.loss.backward()
    finally:
        undo()
    grads = {"grad." + n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    m.eval()
    with torch.no_grad():
        ev = m(**batch)
    assert abs(ev.loss.item() - out.loss.item()) > 1e-4, "dropout was not live"
    npz("G17_train_mode", gen_seed=np.array([1717]), p_hidden=np.array([0.1]), p_att=np.array([0.1]), p_adapter=np.array([0.1]),
        logits=out.logits, loss=out.loss, **{"in." + k: v for k, v in batch.items()}, **grads)


ABLATIONS = [(0, 0, True), (8, 8, False), (0, 8, True), (8, 0, False)]  # (ds_factor_attn, ds_factor_ff, ft_ln)


def g18_ablation_flags(deberta):
    """The reference's ablation switches (args.py:333-337 -> model/deberta.py:252,326: `ds_factor_attn / ds_factor_ff = 0` builds no
    adapter at that site; :1152-1158: `ft_ln=False` leaves the LayerNorms frozen) on the tiny model, eval mode: which parameters
    exist, which are trainable, logits, loss and every trainable gradient -- per variant.  Pins the oracle the GPU test
    test_freeze_policy_flag_variants_vs_oracle compares the HIP path with."""
    from oracle.deberta_oracle import synth_params

    out = {}
    for ds_a, ds_f, ft_ln in ABLATIONS:
        tag = f"a{ds_a}_f{ds_f}_ln{int(ft_ln)}"
        cfg = _tiny_cfg(ds_factor_attn=ds_a, ds_factor_ff=ds_f)
        P = synth_params(cfg, seed=61, std=0.05, ln_jitter=0.1)
        m = build_ref_model(deberta, cfg, P, ft_ln=ft_ln)
        batch = synth_batch(cfg, B=2, L=13, seed=18)
        o = m(**batch)
        o.loss.backward()
        names = sorted(n for n, p in m.named_parameters() if p.requires_grad)
        assert names, tag
        assert any("adapter" in n for n, _ in m.named_parameters()) == bool(ds_a or ds_f), tag
        out[tag + "/trainable"] = np.array("\n".join(names))
        out[tag + "/parameters"] = np.array("\n".join(sorted(n for n, _ in m.named_parameters())))
        out[tag + "/logits_every_4th_column"] = o.logits[..., ::4]  # (fixture size; the loss ties down the rest)
        out[tag + "/loss"] = o.loss
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert p.grad is not None, (tag, n)
                out[tag + "/grad." + n] = p.grad
            else:
                assert p.grad is None, (tag, n)
        if tag == "a0_f0_ln1":
            out.update({"in." + k: v for k, v in batch.items()})
    npz("G18_ablation_flags", **out)


def g5c_attentions(deberta):
    """The tiny model of G5 (same weights, same batch) with output_attentions=True: the reference's `attentions` tuple -- one
    [B, heads, S, S] probability tensor per encoder layer (model/deberta.py:544-560) -- plus logits to tie the two calls."""
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg()
    P = synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    batch = synth_batch(cfg, B=3, L=27, seed=55)
    with torch.no_grad():
        out = m(**batch, output_attentions=True)
    assert len(out.attentions) == cfg.num_hidden_layers
    npz("G5c_tiny_attentions", seed=np.array([5]), batch_seed=np.array([55]), logits=out.logits,
        attentions=torch.stack(out.attentions, 0))


def g6_xlarge(deberta):
    """True xlarge dims, weights from seed recipe (not stored), B=2 S=266 eval: logits slice + checksum + loss."""
    from oracle.deberta_oracle import OracleConfig, synth_params

    cfg = OracleConfig()
    P = synth_params(cfg, seed=0)
    m = build_ref_model(deberta, cfg, P)
    batch = synth_batch(cfg, B=2, L=256, seed=66)
    with torch.no_grad():
        out = m(**batch)
    lg = out.logits
    npz("G6_xlarge", seed=np.array([0]), batch_seed=np.array([66]), loss=out.loss,
        logits_slice=lg[:, ::19, ::997].contiguous(), logits_sum=lg.double().sum(), logits_abs_sum=lg.double().abs().sum(),
        logits_row0=lg[0, 12, :2048].contiguous(), argmax=lg.argmax(-1),
        top5=lg.topk(5, -1).indices[:, ::7].contiguous())


def g7_misc(misc):
    class Tok:
        mask_token = "[MASK]"
        _pad_token = "[PAD]"
        pad_token_id = 0

        def __len__(self):
            return 1000

        def get_special_tokens_mask(self, val, already_has_special_tokens=True):
            return [1 if v in (1, 2) else 0 for v in val]

        def convert_tokens_to_ids(self, t):
            return 4

    out = {}
    for seed in (0, 1):
        g = torch.Generator().manual_seed(700 + seed)
        ids = torch.randint(5, 1000, (4, 33), generator=g)
        ids[:, 0] = 1
        lens = torch.tensor([33, 20, 9, 2])
        for b in range(4):
            ids[b, lens[b] - 1] = 2
            ids[b, lens[b]:] = 0
        torch.manual_seed(seed)
        inp, lab = misc.mask_tokens(ids.clone(), Tok(), 0.15)
        out[f"ids_{seed}"] = ids
        out[f"inputs_{seed}"] = inp
        out[f"labels_{seed}"] = lab
    vl = torch.tensor([10, 3, 0, 7])
    out["video_len"] = vl
    out["get_mask"] = misc.get_mask(vl, 10)

    class A:
        pass

    class Opt:
        param_groups = [{"lr": 0.0}]

    lrs = []
    for sched in ("", "linear_with_warmup"):
        a = A()
        a.lr, a.schedule, a.fraction_warmup_steps = 3e-4, sched, 0.1
        for step in (0, 1, 9, 10, 11, 50, 99, 100):
            o = Opt()
            misc.adjust_learning_rate(o, step, 100, a)
            lrs.append(o.param_groups[0]["lr"])
    out["lrs"] = np.array(lrs, dtype=np.float64)
    npz("G7_misc", **out)


def load_bert():
    """model/bert.py under transformers 5."""
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = None
    orig = mu.PreTrainedModel.get_extended_attention_mask

    def gem(self, mask, shape, device=None, dtype=None):
        # the reference pins transformers 4.17: (1 - mask) * -10000 (v5 would use finfo.min; identical after softmax
        # unless a row is fully masked)
        return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0

    mu.PreTrainedModel.get_extended_attention_mask = gem
    mu.PreTrainedModel.get_head_mask = lambda self, hm, n, *a: [None] * n
    spec = importlib.util.spec_from_file_location("model.bert", os.path.join(REF, "model/bert.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["model.bert"] = mod
    spec.loader.exec_module(mod)
    return mod


def build_ref_bert(bert, cfg, P):
    from transformers import BertConfig

    hc = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                    layer_norm_eps=cfg.layer_norm_eps, pad_token_id=cfg.pad_token_id)
    m = bert.BertForMaskedLM(hc, max_feats=cfg.max_feats, features_dim=cfg.features_dim, freeze_lm=True, ft_ln=True, freeze_mlm=True,
                             n_ans=cfg.n_ans, freeze_last=True)
    sd = m.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    extra = [k for k in sd if k not in P and "position_ids" not in k and "decoder" not in k and "pooler" not in k]
    assert not extra, extra
    m.load_state_dict({k: v.clone() for k, v in P.items()}, strict=False)
    # transformers-4.17 tie_weights semantics (not applied under shim 1)
    m.cls.predictions.decoder.weight = m.bert.embeddings.word_embeddings.weight
    m.cls.predictions.decoder.bias = m.cls.predictions.bias
    m.eval()
    return m


def g8_bert():
    """Config 1 (BERT variant, CPU reference path): tiny config with full tensors + BERT-base dims (seed recipe)."""
    from oracle.bert_oracle import BertOracleConfig, synth_params

    bert = load_bert()
    tiny = BertOracleConfig(vocab_size=211, hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                            max_position_embeddings=64, features_dim=24, max_feats=4)
    P = synth_params(tiny, seed=8, std=0.3, ln_jitter=0.1)
    m = build_ref_bert(bert, tiny, P)
    b = synth_batch(tiny, B=3, L=12, seed=81)
    with torch.no_grad():
        out = m(**b, output_hidden_states=True)
        out_nv = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
    npz("G8_bert_tiny", **{"P/" + k: v for k, v in P.items()}, **{"in/" + k: v for k, v in b.items()},
        logits=out.logits, loss=out.loss, hidden_last=out.hidden_states[-1], hidden_emb=out.hidden_states[0],
        logits_text_only=out_nv.logits, loss_text_only=out_nv.loss)
    # BASELINE configs[0]: BERT-base, B=4, T=10 x 768, L=64, ids ~ U[1000, 30522), all-ones masks
    cfg = BertOracleConfig()
    P = synth_params(cfg, seed=0)
    m = build_ref_bert(bert, cfg, P)
    g = torch.Generator().manual_seed(18)
    video = torch.randn(4, 10, 768, generator=g)
    ids = torch.randint(1000, 30522, (4, 64), generator=g)
    sel = torch.rand(4, 64, generator=g) < 0.15
    sel[:, 1] = True
    labels = torch.where(sel, ids, torch.full_like(ids, -100))
    with torch.no_grad():
        out = m(video=video, video_mask=torch.ones(4, 10, dtype=torch.long), input_ids=ids,
                attention_mask=torch.ones(4, 64, dtype=torch.long), labels=labels)
    lg = out.logits
    npz("G8_bert_base", seed=np.array([0]), batch_seed=np.array([18]), loss=out.loss,
        logits_slice=lg[:, ::7, ::499].contiguous(), logits_sum=lg.double().sum(), logits_abs_sum=lg.double().abs().sum(),
        logits_row0=lg[0, 12, :2048].contiguous(), argmax=lg.argmax(-1), top5=lg.topk(5, -1).indices[:, ::5].contiguous())


def g9_answers(deberta):
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg(n_ans=50)
    P = synth_params(cfg, seed=9, std=0.05, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    g = torch.Generator().manual_seed(99)
    a2tok = torch.randint(5, cfg.vocab_size, (50, 5), generator=g)
    alen = torch.randint(1, 6, (50,), generator=g)
    a2tok = a2tok * (torch.arange(5)[None] < alen[:, None])
    m.set_answer_embeddings(a2tok)
    batch = synth_batch(cfg, B=3, L=27, seed=91)
    batch.pop("labels")
    with torch.no_grad():
        out = m(**batch)
    lg = out.logits
    npz("G9_answers", seed=np.array([9]), a2tok=a2tok, answer_embeddings=m.answer_embeddings.weight,
        answer_bias=m.answer_bias, logits=lg, top10=lg.softmax(-1).topk(10, -1).indices,
        **{"in." + k: v for k, v in batch.items()})


def load_downstream():
    """The reference's videoqa.py / mc.py modules (loops only) with their heavy imports stubbed out."""
    sys.modules.setdefault("hostlist", types.ModuleType("hostlist"))  # util/dist.py imports it at module level
    ds = types.ModuleType("datasets")
    for n in ("build_videoqa_dataset", "videoqa_collate_fn", "build_mc_dataset", "mc_collate_fn",
              "build_videotext_dataset", "videotext_collate_fn"):
        setattr(ds, n, None)
    saved = sys.modules.get("datasets")
    sys.modules["datasets"] = ds
    pkg = sys.modules["model"]
    pkg.build_model = pkg.get_tokenizer = None
    sys.path.insert(0, REF)
    try:
        mods = []
        for name in ("videoqa", "mc", "main"):
            spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mods.append(mod)
    finally:
        sys.path.remove(REF)
        if saved is not None:
            sys.modules["datasets"] = saved
        else:
            del sys.modules["datasets"]
    return mods


def zero_dropout(m):
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def _downstream_model(deberta, n_ans, seed, max_feats=4):
    from oracle.deberta_oracle import synth_params

    cfg = _tiny_cfg(n_ans=n_ans, max_feats=max_feats, vocab_size=300, max_position_embeddings=128)
    P = synth_params(cfg, seed=seed, std=0.08, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    g = torch.Generator().manual_seed(seed + 1)
    a2tok = torch.randint(5, cfg.vocab_size, (n_ans, 3), generator=g)
    alen = torch.randint(1, 4, (n_ans,), generator=g)
    a2tok = a2tok * (torch.arange(3)[None] < alen[:, None])
    m.set_answer_embeddings(a2tok)
    zero_dropout(m)
    return cfg, P, m, a2tok


def _trainable_state(m):
    return {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}


def g10_videoqa(deberta, ref_videoqa):
    """Config 4: the reference's videoqa.py evaluate (msrvtt-style exact match and iVQA soft score) and one epoch of
    its train_one_epoch (dropout 0, Adam) on synthetic batches -- results dict, metrics, loss and updated parameters."""
    import json
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_videoqa_batches

    n_ans = 40
    cfg, P, m, a2tok = _downstream_model(deberta, n_ans, seed=10)
    tok = StubTokenizer(cfg.vocab_size)
    args = Args(max_feats=cfg.max_feats)
    out = {"a2tok": a2tok}
    for name in ("msrvtt", "ivqa"):
        batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_ans, n_batches=3, B=4, seed=101,
                                       dataset_name=name)
        results, metrics = ref_videoqa.evaluate(m, tok, ListLoader(batches), torch.device("cpu"), name, args,
                                                thresholds=[1, 10], split="test", type_map={0: "a", 1: "b"})
        out[f"eval_{name}_results"] = np.array(json.dumps(results))
        out[f"eval_{name}_metrics"] = np.array(json.dumps({k: float(v) for k, v in metrics.items()}))
        # raw mask-row probabilities of the first batch, for tolerance-aware comparison of the top-k ids
        b0 = batches[0]
        enc = tok(b0["text"])
        ids, att = enc["input_ids"].clone(), enc["attention_mask"].clone()
        att[ids == tok.sep_token_id] = 0
        ids[ids == tok.sep_token_id] = tok.pad_token_id
        with torch.no_grad():
            lg = m(video=b0["video"], video_mask=ref_videoqa.get_mask(b0["video_len"], cfg.max_feats), input_ids=ids,
                   attention_mask=att)["logits"]
        out[f"eval_{name}_probs0"] = lg[:, cfg.max_feats:][enc["input_ids"] == tok.mask_token_id].softmax(-1)
    # training: 1 epoch x 3 steps, both loss types
    for name in ("msrvtt", "ivqa"):
        cfg, P, m, a2tok = _downstream_model(deberta, n_ans, seed=10)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
        batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_ans, n_batches=3, B=4, seed=102,
                                       dataset_name=name)
        before = _trainable_state(m)
        stats = ref_videoqa.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device("cpu"), 0, name, args, max_norm=0.1)
        after = _trainable_state(m)
        out[f"train_{name}_stats"] = np.array(json.dumps({k: float(v) for k, v in stats.items()}))
        for key in ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
                    "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight"):
            out[f"train_{name}_delta/{key}"] = after[key] - before[key]
    npz("G10_videoqa", **out)


def g11_mc(deberta, ref_mc):
    """Config 5: the reference's mc.py evaluate (4-way, Yes/No head) and one epoch of its train_one_epoch."""
    import json
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_mc_batches

    cfg, P, m, a2tok = _downstream_model(deberta, 2, seed=11)
    tok = StubTokenizer(cfg.vocab_size)
    args = Args(max_feats=cfg.max_feats)
    out = {"a2tok": a2tok}
    batches = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_choices=4, n_batches=3, B=4, seed=111)
    results, acc = ref_mc.evaluate(m, tok, ListLoader(batches, mc=4), torch.device("cpu"), "how2qa", args)
    out["eval_results"] = np.array(json.dumps(results))
    out["eval_acc"] = np.array([acc])
    hidden = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_choices=4, n_batches=1, B=4, seed=112, with_gt=False)
    results, acc = ref_mc.evaluate(m, tok, ListLoader(hidden, mc=4), torch.device("cpu"), "how2qa", args)
    out["eval_hidden_results"] = np.array(json.dumps(results))
    # candidate scores of the first batch
    b0 = batches[0]
    sc = []
    with torch.no_grad():
        for aid in range(4):
            enc = tok(b0["text"][aid])
            lg = m(video=b0["video"], video_mask=ref_mc.get_mask(b0["video_len"], cfg.max_feats), input_ids=enc["input_ids"],
                   attention_mask=enc["attention_mask"])["logits"]
            sc.append(lg[:, cfg.max_feats:][enc["input_ids"] == tok.mask_token_id].softmax(-1)[:, 0])
    out["eval_scores0"] = torch.stack(sc, 1)
    cfg, P, m, a2tok = _downstream_model(deberta, 2, seed=11)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    tb = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_choices=4, n_batches=3, B=4, seed=113)
    before = _trainable_state(m)
    stats = ref_mc.train_one_epoch(m, tok, ListLoader(tb, mc=4), opt, torch.device("cpu"), 0, args, max_norm=0.1)
    after = _trainable_state(m)
    out["train_stats"] = np.array(json.dumps({k: float(v) for k, v in stats.items()}))
    for key in ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
                "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight"):
        out[f"train_delta/{key}"] = after[key] - before[key]
    npz("G11_mc", **out)


def write_feature_fixture(root, seed=12):
    """Synthetic WebVid-style inputs: csv (video_id, text) + one fp16 .npy [n_sec, F] per video (one missing, one
    corrupt).  Deterministic; used by the golden generator and regenerated identically by the tests."""
    os.makedirs(os.path.join(root, "feats"), exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    lens = [3, 10, 11, 25, 1, 47, 10, 7, 0]
    rows = []
    for i, n in enumerate(lens):
        vid = 1000 + i
        rows.append((vid, f"caption number {i}, with a comma"))
        if i == 7:
            continue  # missing file
        path = os.path.join(root, "feats", f"{vid}.mp4.npy")
        if n == 0:
            with open(path, "wb") as f:
                f.write(b"not a npy file")  # corrupt file
            continue
        np.save(path, torch.randn(n, 16, generator=g).half().numpy())
    import csv as _csv

    with open(os.path.join(root, "data.csv"), "w", newline="") as f:
        w = _csv.writer(f)
        w.writerow(["video_id", "text"])
        w.writerows(rows)
    return os.path.join(root, "data.csv"), os.path.join(root, "feats")


def g12_dataset():
    """datasets/videotext_dataset.py on the synthetic feature files: items and one collated batch."""
    import tempfile

    spec = importlib.util.spec_from_file_location("ref_videotext_dataset", os.path.join(REF, "datasets/videotext_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as d:
        csv_path, feats = write_feature_fixture(d)
        ds = mod.VideoText_Dataset(csv_path, feats, max_feats=10, features_dim=16)
        items = [ds[i] for i in range(len(ds))]
        batch = mod.videotext_collate_fn(items)
    npz("G12_dataset", video=batch["video"], video_len=batch["video_len"], text=np.array(batch["text"]))


def g13_main_loops(deberta, ref_main):
    """main.py train_one_epoch / evaluate (MLM) of the reference on synthetic caption batches: seeded CPU mask_tokens."""
    import json
    from oracle.deberta_oracle import synth_params
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_videotext_batches

    cfg = _tiny_cfg(max_feats=4, vocab_size=300, max_position_embeddings=128)
    P = synth_params(cfg, seed=21, std=0.08, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    zero_dropout(m)
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=31)
    torch.manual_seed(7)
    ev = ref_main.evaluate(m, tok, ListLoader(batches), torch.device("cpu"), args)
    torch.manual_seed(8)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    before = _trainable_state(m)
    tr = ref_main.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device("cpu"), 0, args, 0.1)
    after = _trainable_state(m)
    out = {"eval_stats": np.array(json.dumps({k: float(v) for k, v in ev.items()})),
           "train_stats": np.array(json.dumps({k: float(v) for k, v in tr.items()}))}
    for key in ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
                "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight"):
        out[f"train_delta/{key}"] = after[key] - before[key]
    npz("G13_main_loops", **out)


XL_GRAD_FULL = ("deberta.embeddings.linear_video.bias", "deberta.encoder.LayerNorm.weight", "deberta.encoder.LayerNorm.bias",
                "deberta.embeddings.LayerNorm.weight", "deberta.embeddings.LayerNorm.bias",
                "lm_predictions.lm_head.LayerNorm.weight", "lm_predictions.lm_head.LayerNorm.bias")
XL_GRAD_LAYERS = (0, 12, 23)


def xl_grad_slices(name, g):
    """What G6b keeps of one xlarge gradient tensor: small tensors whole, matrices as a strided slice."""
    if g.dim() == 1:
        return g
    return g[::4, ::8].contiguous()


def g6b_xlarge_backward(deberta):
    """True xlarge dims (24 layers, H=1536), eval-mode math, B=2, S=266: the reference's own BACKWARD.  Stored: the
    Frobenius norm of every trainable gradient (298 tensors), full vectors / strided slices of linear_video, the three
    stand-alone LayerNorms and everything trainable in layers 0, 12 and 23, loss and a logits slice."""
    from oracle.deberta_oracle import OracleConfig, synth_params

    cfg = OracleConfig()
    P = synth_params(cfg, seed=0)
    m = build_ref_model(deberta, cfg, P)
    batch = synth_batch(cfg, B=2, L=256, seed=67)
    out = m(**batch)
    out.loss.backward()
    names, norms, keep = [], [], {}
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, n
        names.append(n)
        norms.append(float(p.grad.double().norm()))
        layer_hit = any(n.startswith(f"deberta.encoder.layer.{i}.") for i in XL_GRAD_LAYERS)
        if n in XL_GRAD_FULL or layer_hit or "linear_video.weight" in n or n.startswith("deberta.encoder.conv."):
            keep["grad." + n] = xl_grad_slices(n, p.grad)
    lg = out.logits.detach()
    npz("G6b_xlarge_backward", seed=np.array([0]), batch_seed=np.array([67]), loss=out.loss.detach(),
        logits_slice=lg[:, ::19, ::997].contiguous(), names=np.array(names), norms=np.array(norms, dtype=np.float64), **keep)


def g14_g15_xlarge_downstream(deberta):
    """True xlarge dims for the two downstream configurations (eval forward only):
    G14 = cfg 5 (mc.py): S = 512 (T=10 + L=502), Yes/No answer head (n_ans=2), one candidate text per sample, B=2;
    G15 = cfg 4 (videoqa.py): n_ans = 1000 answer table from set_answer_embeddings, one [MASK] per row, top-10 ids."""
    from oracle.deberta_oracle import OracleConfig, synth_params

    cfg = OracleConfig(n_ans=1000)
    P = synth_params(cfg, seed=0)
    m = build_ref_model(deberta, cfg, P)
    g = torch.Generator().manual_seed(150)
    a2tok = torch.randint(5, cfg.vocab_size, (1000, 5), generator=g)
    alen = torch.randint(1, 6, (1000,), generator=g)
    a2tok = a2tok * (torch.arange(5)[None] < alen[:, None])
    m.set_answer_embeddings(a2tok)
    MASK = 128000  # [MASK] of the deberta-v2 vocabulary
    batch = synth_batch(cfg, B=2, L=256, seed=151)
    batch.pop("labels")
    tlen = batch["attention_mask"].sum(1)
    mpos = torch.stack([torch.randint(1, int(t), (1,), generator=g) for t in tlen]).view(-1)
    batch["input_ids"][torch.arange(2), mpos] = MASK
    with torch.no_grad():
        lg = m(**batch)["logits"]
    rows = lg[:, cfg.max_feats:][batch["input_ids"] == MASK]  # [2, 1000]   (videoqa.py:164-168)
    probs = rows.softmax(-1)
    npz("G15_xlarge_videoqa", seed=np.array([0]), batch_seed=np.array([151]), a2tok_seed=np.array([150]), mask_id=np.array([MASK]),
        mask_pos=mpos, mask_logits=rows, mask_probs=probs, top10=probs.topk(10, -1).indices, top10_probs=probs.topk(10, -1).values,
        logits_slice=lg[:, ::19, ::97].contiguous())
    # ---- cfg 5: the same model re-headed to the 2-answer table (mc.py:281-305), S = 512
    a2 = torch.tensor([[2748, 0], [1302, 0]])
    m.set_answer_embeddings(a2)
    b5 = synth_batch(cfg, B=2, L=502, seed=141)
    b5.pop("labels")
    tlen = b5["attention_mask"].sum(1)
    mpos5 = torch.stack([torch.randint(1, int(t), (1,), generator=g) for t in tlen]).view(-1)
    b5["input_ids"][torch.arange(2), mpos5] = MASK
    with torch.no_grad():
        lg5 = m(**b5)["logits"]  # [2, 512, 2]
    score = lg5[:, cfg.max_feats:][b5["input_ids"] == MASK].softmax(-1)[:, 0]  # mc.py:166-172
    npz("G14_xlarge_mc", seed=np.array([0]), batch_seed=np.array([141]), a2tok=a2, mask_id=np.array([MASK]), mask_pos=mpos5,
        logits=lg5, score=score)


def g16_checkpoint(deberta, ref_main):
    """A checkpoint WRITTEN BY THE REFERENCE: main.py's epoch (train_one_epoch with torch.optim.Adam) followed by the
    save_on_master(...) call of main.py:290-300 through the reference's own util/dist.py, on the tiny model.  Also stored:
    the trainable-parameter order (the index space of the optimizer state), what the reference computes after loading the
    file again (eval loss) and what one more resumed training epoch does to a few parameters."""
    import argparse as _ap
    import json
    from oracle.deberta_oracle import synth_params
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_videotext_batches

    spec = importlib.util.spec_from_file_location("ref_util_dist", os.path.join(REF, "util/dist.py"))
    rdist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rdist)
    cfg = _tiny_cfg(max_feats=4, vocab_size=300, max_position_embeddings=128)
    P = synth_params(cfg, seed=22, std=0.08, ln_jitter=0.1)
    m = build_ref_model(deberta, cfg, P)
    zero_dropout(m)
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=32)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    torch.manual_seed(9)
    ref_main.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device("cpu"), 0, args, 0.1)
    path = os.path.join(OUT, "G16_reference_checkpoint.pth")
    ns = _ap.Namespace(lr=1e-3, epochs=2, beta1=0.9, beta2=0.95, clip_max_norm=0.1)
    rdist.save_on_master({"model": m.state_dict(), "optimizer": opt.state_dict(), "epoch": 0, "args": ns}, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KB)")
    order = [n for n, p in m.named_parameters() if p.requires_grad]
    # what the reference does with the file (main.py:235-243): a fresh model + optimizer, load, evaluate, resume one epoch
    m2 = build_ref_model(deberta, cfg, synth_params(cfg, seed=23, std=0.08))
    zero_dropout(m2)
    opt2 = torch.optim.Adam([p for p in m2.parameters() if p.requires_grad], lr=5e-4, betas=(0.9, 0.95))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    m2.load_state_dict(ck["model"], strict=False)
    opt2.load_state_dict(ck["optimizer"])
    torch.manual_seed(10)
    ev = ref_main.evaluate(m2, tok, ListLoader(batches), torch.device("cpu"), args)
    before = _trainable_state(m2)
    torch.manual_seed(11)
    m2.train()
    tr = ref_main.train_one_epoch(m2, tok, ListLoader(batches), opt2, torch.device("cpu"), 1, args, 0.1)
    after = _trainable_state(m2)
    out = {"trainable_order": np.array(order), "eval_stats": np.array(json.dumps({k: float(v) for k, v in ev.items()})),
           "resume_train_stats": np.array(json.dumps({k: float(v) for k, v in tr.items()}))}
    for key in ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
                "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight"):
        out[f"resume_delta/{key}"] = after[key] - before[key]
    npz("G16_checkpoint_meta", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-xl", action="store_true")
    args = ap.parse_args()
    adapter_mod, deberta, misc = load_reference()
    torch.set_grad_enabled(True)
    jobs = {
        "G1": lambda: g1_adapter(adapter_mod),
        "G2": lambda: g2_relpos(deberta),
        "G3": lambda: g3_attention(deberta),
        "G3b": lambda: g3b_attention_backward(deberta),
        "G4": lambda: g4_layer_conv(deberta),
        "G5": lambda: g5_tiny_model(deberta),
        "G5c": lambda: g5c_attentions(deberta),
        "G6": lambda: g6_xlarge(deberta),
        "G7": lambda: g7_misc(misc),
        "G8": g8_bert,
        "G9": lambda: g9_answers(deberta),
        "G10": lambda: g10_videoqa(deberta, load_downstream()[0]),
        "G11": lambda: g11_mc(deberta, load_downstream()[1]),
        "G12": g12_dataset,
        "G13": lambda: g13_main_loops(deberta, load_downstream()[2]),
        "G6b": lambda: g6b_xlarge_backward(deberta),
        "G14": lambda: g14_g15_xlarge_downstream(deberta),  # writes G14 and G15
        "G16": lambda: g16_checkpoint(deberta, load_downstream()[2]),
        "G17": lambda: g17_train_mode(deberta, adapter_mod),
        "G18": lambda: g18_ablation_flags(deberta),
    }
    for k, fn in jobs.items():
        if args.only and k not in args.only.split(","):
            continue
        if k in ("G6", "G6b", "G14") and args.skip_xl:
            continue
        fn()


if __name__ == "__main__":
    main()
