"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference (tests/golden/).

CPU only.  Tolerances: integer tables bit-exact; fp32 activations <= 2e-5 max-abs (same ATen kernels,
different association order); grads <= 1e-4 relative to their max.
"""
import numpy as np
import pytest
import torch

from oracle import deberta_oracle as O
from oracle import misc_oracle as MO
from tests.golden.make_goldens import _tiny_cfg, synth_batch


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


def test_g1_adapter(golden):
    g = golden("G1_adapter")
    P = {"a." + k[2:]: v for k, v in g.items() if k.startswith("w.")}
    y = O.adapter(g["x"], P, "a")
    assert maxabs(y, g["y"]) < 1e-6


@pytest.mark.parametrize("S", [1, 2, 74, 129, 266, 512])
def test_g2_relpos_bit_exact(golden, S):
    g = golden("G2_relpos")
    r = O.relative_position(S, S, 256, 512)
    assert r.dtype == np.int64
    assert (r[:, 0] == g[f"col_{S}"].numpy()).all()
    assert (r[0, :] == g[f"row_{S}"].numpy()).all()
    chk = np.array([int(np.abs(r).sum()), int((r * np.arange(S)[None, :]).sum())])
    assert (chk == g[f"sum_{S}"].numpy()).all()
    # antisymmetry => the p2c index equals the c2p index; Toeplitz vector form matches the table
    cfg = O.OracleConfig()
    idx = O.rel_index_by_delta(S, cfg)
    i, j = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    c2p = np.clip(r + 256, 0, 511)
    p2c_t = np.clip(-r.T + 256, 0, 511)
    assert (c2p == p2c_t).all()
    assert (idx[i - j + S - 1] == c2p).all()


@pytest.mark.parametrize("S", [37, 266])
def test_g3_attention(golden, S):
    g = golden("G3_attention")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
    hidden, qs, mask = g[f"hidden_{S}"], g[f"qs_{S}"], g[f"mask_{S}"]
    m = mask.float()
    mask4d = (m[:, None, None, :] * m[:, None, :, None]).to(torch.uint8)
    rel = O.relative_position(S, S, 256, 512)
    remb = O._ln(P["deberta.encoder.rel_embeddings.weight"], P, "deberta.encoder.LayerNorm", cfg.layer_norm_eps)
    assert maxabs(remb, g["rel_emb"]) < 1e-5
    pre = "deberta.encoder.layer.1.attention.self"
    y0 = O.disentangled_attention(hidden, mask4d, rel, remb, P, pre, cfg)
    y1 = O.disentangled_attention(hidden, mask4d, rel, remb, P, pre, cfg, query_states=qs)
    assert maxabs(y0, g[f"ctx_{S}"]) < 2e-5
    assert maxabs(y1, g[f"ctxq_{S}"]) < 2e-5
    # padded query rows are exactly zero (XSoftmax turns the all -inf row into 0)
    assert (y0[0, S - 5:] == 0).all()


@pytest.mark.parametrize("S", [37, 266])
def test_g3b_attention_backward(golden, S):
    """autograd of the oracle's attention against the reference's own backward of DisentangledSelfAttention (G3b): input
    gradients w.r.t. hidden_states, query_states and the (LayerNorm-ed) relative-position table, for a seeded upstream dy"""
    g, gb = golden("G3_attention"), golden("G3b_attention_backward")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
    hidden, qs, mask, dy = g[f"hidden_{S}"], g[f"qs_{S}"], g[f"mask_{S}"], gb[f"dy_{S}"]
    m = mask.float()
    mask4d = (m[:, None, None, :] * m[:, None, :, None]).to(torch.uint8)
    rel = O.relative_position(S, S, 256, 512)
    pre = "deberta.encoder.layer.1.attention.self"
    sl = (lambda t: t) if S == 37 else (lambda t: t[:, ::3])
    tol = 2e-5 if S == 37 else 2e-3  # S=266 is stored in fp16
    for tag, use_q in (("", False), ("q", True)):
        h = hidden.clone().requires_grad_(True)
        q = qs.clone().requires_grad_(True) if use_q else None
        remb = g["rel_emb"].clone().requires_grad_(True)
        y = O.disentangled_attention(h, mask4d, rel, remb, P, pre, cfg, query_states=q)
        y.backward(dy)
        scale = lambda ref: max(1.0, ref.abs().max().item())
        for got, key in ((sl(h.grad), f"dhidden{tag}_{S}"), (remb.grad, f"drel{tag}_{S}")) + (((sl(q.grad), f"dqs_{S}"),) if use_q else ()):
            ref = gb[key].float()
            assert maxabs(got, ref) < tol * scale(ref), (key, maxabs(got, ref))


def test_g5c_attention_probabilities(golden):
    """the oracle's attention probabilities per encoder layer against the reference's `attentions` tuple (G5c), on the
    hidden states the reference itself produced for the same forward (G5)"""
    g5, gc = golden("G5_tiny_model"), golden("G5c_tiny_attentions")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    mask = torch.cat([g5["in.video_mask"], g5["in.attention_mask"]], 1).float()
    S = mask.shape[1]
    mask4d = (mask[:, None, None, :] * mask[:, None, :, None]).to(torch.uint8)
    rel = O.relative_position(S, S, 256, 512)
    remb = O._ln(P["deberta.encoder.rel_embeddings.weight"], P, "deberta.encoder.LayerNorm", cfg.layer_norm_eps)
    assert gc["attentions"].shape[0] == cfg.num_hidden_layers
    for li in range(cfg.num_hidden_layers):
        _, probs = O.disentangled_attention(g5["hidden_states"][li], mask4d, rel, remb, P,
                                            f"deberta.encoder.layer.{li}.attention.self", cfg, return_probs=True)
        assert maxabs(probs, gc["attentions"][li]) < 2e-5, li


def test_g4_layer_conv(golden):
    g = golden("G4_layer_conv")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=4, std=0.05, ln_jitter=0.1)
    hidden, qs, mask = g["hidden"], g["qs"], g["mask"]
    S = hidden.shape[1]
    m = mask.float()
    mask4d = (m[:, None, None, :] * m[:, None, :, None]).to(torch.uint8)
    rel = O.relative_position(S, S, 256, 512)
    remb = O._ln(P["deberta.encoder.rel_embeddings.weight"], P, "deberta.encoder.LayerNorm", cfg.layer_norm_eps)
    pre = "deberta.encoder.layer.2"
    assert maxabs(O.layer(hidden, mask4d, rel, remb, P, pre, cfg), g["y_layer"]) < 2e-5
    assert maxabs(O.layer(hidden, mask4d, rel, remb, P, pre, cfg, query_states=qs), g["y_layer_q"]) < 2e-5
    assert maxabs(O.conv_layer(hidden, qs, mask, P, cfg), g["y_conv"]) < 2e-5


def test_g5_tiny_model_logits_loss_grads(golden):
    g = golden("G5_tiny_model")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    regen = synth_batch(cfg, B=3, L=27, seed=55)
    for k in batch:
        assert torch.equal(batch[k], regen[k]), k  # the seeded generator recipe is reproducible
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    out = O.forward(P, cfg, return_hidden=True, **batch)
    assert maxabs(out["logits"], g["logits"]) < 5e-5
    assert abs(out["loss"].item() - g["loss"].item()) < 1e-5
    hs = torch.stack(out["hidden_states"], 0)
    assert maxabs(hs, g["hidden_states"]) < 2e-5
    out["loss"].backward()
    n = 0
    for k, v in P.items():
        if O.is_trainable(k):
            ref = g["grad." + k]
            tol = 1e-4 * max(ref.abs().max().item(), 1e-3)
            assert maxabs(v.grad, ref) < tol, k
            n += 1
    assert n == len([k for k in g if k.startswith("grad.")])
    # skipping the dead in-encoder pass of the last layer changes neither logits nor loss
    out2 = O.forward(P, cfg, return_hidden=False, **batch)
    assert torch.equal(out2["logits"], out["logits"])


def test_g17_train_mode_with_the_references_dropout_draws(golden):
    """G17: the reference in train() mode with every dropout decision drawn from one seeded generator.  The oracle's dropout
    sites (oracle.dropout_masks), fed masks from an identically seeded generator in the oracle's own execution order, must
    reproduce the reference's loss, logits and every trainable gradient: that pins place, scale and ORDER of all dropout
    sites of the train-mode restatement (model/deberta.py:142-240, 258, 332, 403, 779, 796, 1054; model/adapter.py:40-41)."""
    g = golden("G17_train_mode")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    gen = torch.Generator().manual_seed(int(g["gen_seed"][0]))
    p_of = {"emb": float(g["p_hidden"][0]), "pos": float(g["p_hidden"][0]), "hid": float(g["p_hidden"][0]),
            "conv": float(g["p_hidden"][0]), "att": float(g["p_att"][0]), "ad": float(g["p_adapter"][0])}
    kinds = []

    def provider(kind, shape):
        kinds.append(kind)
        p = p_of[kind]
        return torch.empty(shape).bernoulli_(1 - p, generator=gen) / (1 - p)

    with O.dropout_masks(provider):
        out = O.forward(P, cfg, return_hidden=True, **batch)  # (the reference runs -- and draws for -- the dead last-layer pass)
    nexec = cfg.num_hidden_layers + 2
    assert kinds.count("pos") == kinds.count("att") == nexec and kinds.count("ad") == kinds.count("hid") == 2 * nexec
    assert kinds.count("emb") == kinds.count("conv") == 1 and kinds[0] == "emb"
    assert abs(out["loss"].item() - g["loss"].item()) < 1e-5
    assert maxabs(out["logits"], g["logits"]) < 5e-5
    out["loss"].backward()
    n = 0
    for k, v in P.items():
        if O.is_trainable(k):
            ref = g["grad." + k]
            assert maxabs(v.grad, ref) < 1e-4 * max(ref.abs().max().item(), 1e-3), k
            n += 1
    assert n == len([k for k in g if k.startswith("grad.")])
    # without masks the same call is the eval-mode forward of G5
    g5 = golden("G5_tiny_model")
    assert maxabs(O.forward(P, cfg, **batch)["logits"], g5["logits"]) < 5e-5


def test_g5b_text_only(golden):
    g5 = golden("G5_tiny_model")
    gb = golden("G5b_tiny_textonly")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    with torch.no_grad():
        out = O.forward(P, cfg, g5["in.input_ids"], g5["in.attention_mask"])
    assert maxabs(out["logits"], gb["logits"]) < 5e-5


def test_g7_misc_bit_exact(golden):
    g = golden("G7_misc")
    assert torch.equal(MO.get_mask(g["video_len"], 10), g["get_mask"])
    assert MO.get_mask(g["video_len"], 10).dtype == torch.int64
    for seed in (0, 1):
        ids = g[f"ids_{seed}"].clone()
        special = (ids == 1) | (ids == 2)
        torch.manual_seed(seed)
        inp, lab = MO.mask_tokens(ids, special, 0, 4, 1000, 0.15)
        assert torch.equal(inp, g[f"inputs_{seed}"])
        assert torch.equal(lab, g[f"labels_{seed}"])
    lrs = []
    for sched in ("", "linear_with_warmup"):
        for step in (0, 1, 9, 10, 11, 50, 99, 100):
            lrs.append(MO.lr_at(step, 100, 3e-4, sched, 0.1))
    assert np.array_equal(np.array(lrs), g["lrs"].numpy())


def test_g9_answer_head(golden):
    g = golden("G9_answers")
    cfg = _tiny_cfg(n_ans=50)
    P = O.synth_params(cfg, seed=9, std=0.05, ln_jitter=0.1)
    emb = O.answer_embeddings(g["a2tok"], P, cfg)
    assert maxabs(emb, g["answer_embeddings"]) < 1e-6
    P["answer_embeddings.weight"] = emb
    # reference assigns `answer_bias.weight = ...` (an attribute), the effective bias stays as initialised
    P["answer_bias"] = g["answer_bias"]
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    with torch.no_grad():
        out = O.forward(P, cfg, **batch)
    assert maxabs(out["logits"], g["logits"]) < 5e-5
    top10 = out["logits"].softmax(-1).topk(10, -1).indices
    assert torch.equal(top10, g["top10"])


# ---------------------------------------------------------------------------------- config 1: BERT variant (row a25)
def test_g8_bert_tiny(golden):
    from oracle import bert_oracle as BO

    g = golden("G8_bert_tiny")
    P = {k[2:]: v for k, v in g.items() if k.startswith("P/")}
    b = {k[3:]: v for k, v in g.items() if k.startswith("in/")}
    cfg = BO.BertOracleConfig(vocab_size=211, hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                              max_position_embeddings=64, features_dim=24, max_feats=4)
    assert set(P) == set(BO.param_shapes(cfg)), "state_dict key map"
    emb = BO.embeddings(cfg, P, b["input_ids"], b["video"])
    assert maxabs(emb, g["hidden_emb"]) < 2e-5
    out = BO.forward(cfg, P, b["input_ids"], b["attention_mask"], b["video"], b["video_mask"], b["labels"])
    assert maxabs(out["hidden"], g["hidden_last"]) < 5e-5
    assert maxabs(out["logits"], g["logits"]) < 1e-4
    assert abs(out["loss"].item() - g["loss"].item()) < 1e-5
    assert torch.equal(out["logits"].argmax(-1), g["logits"].argmax(-1))
    out = BO.forward(cfg, P, b["input_ids"], b["attention_mask"], labels=b["labels"])
    assert maxabs(out["logits"], g["logits_text_only"]) < 1e-4
    assert abs(out["loss"].item() - g["loss_text_only"].item()) < 1e-5


@pytest.mark.slow
def test_g8_bert_base_config1(golden):
    """BASELINE configs[0]: BERT-base, 4 synthetic videos (T=10 x 768), L=64, MLM forward on the CPU path."""
    from oracle import bert_oracle as BO

    g = golden("G8_bert_base")
    cfg = BO.BertOracleConfig()
    P = BO.synth_params(cfg, seed=int(g["seed"][0]))
    gen = torch.Generator().manual_seed(int(g["batch_seed"][0]))
    video = torch.randn(4, 10, 768, generator=gen)
    ids = torch.randint(1000, 30522, (4, 64), generator=gen)
    sel = torch.rand(4, 64, generator=gen) < 0.15
    sel[:, 1] = True
    labels = torch.where(sel, ids, torch.full_like(ids, -100))
    with torch.no_grad():
        out = BO.forward(cfg, P, ids, torch.ones(4, 64, dtype=torch.long), video, torch.ones(4, 10, dtype=torch.long), labels)
    lg = out["logits"]
    assert lg.shape == (4, 74, 30522)
    assert maxabs(lg[:, ::7, ::499], g["logits_slice"]) < 1e-4
    assert maxabs(lg[0, 12, :2048], g["logits_row0"]) < 1e-4
    assert abs(lg.double().sum().item() - g["logits_sum"].item()) < 1e-5 * g["logits_abs_sum"].item()
    assert abs(out["loss"].item() - g["loss"].item()) < 1e-5
    assert torch.equal(lg.argmax(-1), g["argmax"])  # token indices bit-exact
    assert torch.equal(lg.topk(5, -1).indices[:, ::5], g["top5"])


def test_bf16_operand_mode_is_opt_in_and_close():
    """oracle.bf16_operands (the HIP path's arithmetic contract for tight gradient checks) changes nothing unless
    entered, and inside it the outputs move by bf16 rounding only."""
    from tests.golden.make_goldens import _tiny_cfg, synth_batch

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    batch = synth_batch(cfg, B=3, L=27, seed=55)
    with torch.no_grad():
        a = O.forward(P, cfg, **batch)
        with O.bf16_operands():
            b = O.forward(P, cfg, **batch)
        c = O.forward(P, cfg, **batch)
    assert torch.equal(a["logits"], c["logits"])
    err = (a["logits"] - b["logits"]).abs().max().item()
    assert 0 < err < 5e-2, err


@pytest.mark.parametrize("ds_a,ds_f,ft_ln", [(0, 0, True), (8, 8, False), (0, 8, True), (8, 0, False)],
                         ids=["no-adapters", "ft_ln-off", "ffn-adapter-only", "attn-adapter-only+ft_ln-off"])
def test_g18_ablation_flags_of_the_reference(golden, ds_a, ds_f, ft_ln):
    """G18: the reference built with its ablation switches (args.py:333-337: ds_factor_attn / ds_factor_ff = 0 -> no adapter at that
    site, model/deberta.py:252,326; ft_ln=False -> LayerNorms stay frozen, :1152-1158).  The oracle must agree on which parameters
    exist, which are trainable (`is_trainable(name, ft_ln)` = the reference's substring rule), on logits, loss and every trainable
    gradient -- it is what the GPU test of the same flags (test_freeze_policy_flag_variants_vs_oracle) compares the HIP path with."""
    g = golden("G18_ablation_flags", raw=True)
    tag = f"a{ds_a}_f{ds_f}_ln{int(ft_ln)}"
    cfg = _tiny_cfg(ds_factor_attn=ds_a, ds_factor_ff=ds_f)
    P = O.synth_params(cfg, seed=61, std=0.05, ln_jitter=0.1)
    ref_params = str(g[tag + "/parameters"]).split("\n")
    ref_train = str(g[tag + "/trainable"]).split("\n")
    # the reference's decoder.weight is an untied copy under the import shims (SURVEY App. B note 6); everything else by name
    assert sorted(k for k in P) == sorted(n for n in ref_params if "decoder" not in n)
    assert sorted(k for k in P if O.is_trainable(k, ft_ln=ft_ln)) == ref_train
    assert any("adapter" in k for k in P) == bool(ds_a or ds_f)
    assert any("LayerNorm" in k for k in ref_train) == ft_ln
    batch = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("in.")}
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k, ft_ln=ft_ln))
    out = O.forward(P, cfg, **batch)
    assert abs(out["loss"].item() - float(g[tag + "/loss"])) < 1e-5
    assert maxabs(out["logits"][..., ::4], torch.from_numpy(g[tag + "/logits_every_4th_column"])) < 5e-5
    out["loss"].backward()
    n = 0
    for k, v in P.items():
        if O.is_trainable(k, ft_ln=ft_ln):
            ref = torch.from_numpy(g[tag + "/grad." + k])
            assert maxabs(v.grad, ref) < 1e-4 * max(ref.abs().max().item(), 1e-3), k
            n += 1
        else:
            assert v.grad is None, k
    assert n == len(ref_train) == len([k for k in g if k.startswith(tag + "/grad.")])
