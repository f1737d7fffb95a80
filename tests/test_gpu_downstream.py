"""Downstream paths on a real MI355X (BASELINE configs 4 and 5): the product's videoqa / mc loops
running the HIP model, against what the REFERENCE's loops returned (goldens G10 / G11), plus the gradient that flows
through the returned logits against the oracle's autograd.

Tolerances: bf16 MFMA operands -> answer probabilities within 3e-2 relative; predicted ids must agree wherever the
reference separates the candidates by more than that, accuracies may move by at most one near-tie flip.
"""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu

from frozenbilm_amd import mc as P_mc  # noqa: E402
from frozenbilm_amd import videoqa as P_vqa  # noqa: E402
from oracle import deberta_oracle as O  # noqa: E402
from oracle.model_wrapper import OracleModel  # noqa: E402
from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_mc_batches, make_videoqa_batches  # noqa: E402
from tests.test_downstream_loops import DELTA_KEYS, N_ANS, _j, cosine, mask_probs, same_ranking, tiny  # noqa: E402

DEV = "cuda"
REL = 3e-2


def hip_model(n_ans, seed, a2tok, train=False):
    from frozenbilm_amd.model.config import DebertaV2Config
    from frozenbilm_amd.model.deberta import DebertaV2ForMaskedLM

    cfg = tiny(n_ans)
    P = O.synth_params(cfg, seed=seed, std=0.08, ln_jitter=0.1)
    c = DebertaV2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        max_position_embeddings=cfg.max_position_embeddings, position_buckets=cfg.position_buckets,
                        layer_norm_eps=cfg.layer_norm_eps, conv_kernel_size=cfg.conv_kernel_size,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = DebertaV2ForMaskedLM(c, max_feats=cfg.max_feats, features_dim=cfg.features_dim, ds_factor_attn=cfg.ds_factor_attn,
                             ds_factor_ff=cfg.ds_factor_ff, n_ans=cfg.n_ans, dropout=0.0)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    m.to(DEV)
    m.set_answer_embeddings(torch.as_tensor(a2tok))
    m.train(train)
    return cfg, P, m


def test_gradient_through_logits_vs_oracle(golden):
    """videoqa.py:66-83: the loss is computed by the caller on output['logits'] -- the logits must be differentiable."""
    g = golden("G10_videoqa", raw=True)
    cfg, P, m = hip_model(N_ANS, 10, g["a2tok"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    b = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 1, 6, seed=77)[0]
    enc = tok(b["text"])
    from frozenbilm_amd.util.misc import get_mask

    vm = get_mask(b["video_len"], cfg.max_feats)
    om = OracleModel(tiny(N_ANS), P, torch.as_tensor(g["a2tok"]))
    lo = P_vqa.mask_row_logits(om(video=b["video"], video_mask=vm, input_ids=enc["input_ids"],
                                  attention_mask=enc["attention_mask"])["logits"], enc["input_ids"], tok, args)
    P_vqa.vqa_loss(lo, b["answer_id"], "msrvtt").backward()
    out = m(video=b["video"].to(DEV), video_mask=vm.to(DEV), input_ids=enc["input_ids"].to(DEV),
            attention_mask=enc["attention_mask"].to(DEV))
    assert out["loss"] is None and out["logits"].requires_grad
    lg = P_vqa.mask_row_logits(out["logits"], enc["input_ids"], tok, args)
    assert (lg.detach().cpu() - lo.detach()).abs().max().item() < 5e-2
    loss = P_vqa.vqa_loss(lg, b["answer_id"].to(DEV), "msrvtt")
    assert abs(loss.item() - P_vqa.vqa_loss(lo, b["answer_id"], "msrvtt").item()) < 2e-2
    loss.backward()
    ref = {n: p.grad for n, p in om.named_ref_parameters().items() if p.requires_grad}
    bad, n = [], 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        n += 1
        assert p.grad is not None, name
        r = ref[name]
        rel = (p.grad.cpu() - r).norm().item() / (r.norm().item() + 1e-12)
        lim = 0.25 if "adapter.down" in name else 6e-2  # bf16-vs-fp32 ReLU gate flips (see test_gpu_model)
        if rel > lim:
            bad.append((name, rel))
    assert n == len(ref) and not bad, bad[:8]


@pytest.mark.parametrize("name", ["msrvtt", "ivqa"])
def test_videoqa_evaluate_gpu(golden, name):
    g = golden("G10_videoqa", raw=True)
    cfg, P, m = hip_model(N_ANS, 10, g["a2tok"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 3, 4, seed=101, dataset_name=name)
    results, metrics = P_vqa.evaluate(m, tok, ListLoader(batches), torch.device(DEV), name, args, thresholds=[1, 10],
                                      split="test", type_map={0: "a", 1: "b"})
    ref_results, ref_metrics = _j(g, f"eval_{name}_results"), _j(g, f"eval_{name}_metrics")
    probs = mask_probs(m, tok, batches, args, dev=DEV)
    p0 = torch.as_tensor(g[f"eval_{name}_probs0"])
    assert ((probs["q0"] - p0[0]).abs() / p0[0]).max().item() < REL
    for q in results:
        assert same_ranking(results[q]["pred"], ref_results[q]["pred"], probs[q], 2 * REL), (q, results[q]["pred"], ref_results[q]["pred"])
    n = len(results)
    for k in ("acc1", "acc10"):
        assert abs(metrics[k] - ref_metrics[k]) <= 1.0 / n + 1e-9, (k, metrics[k], ref_metrics[k])


@pytest.mark.parametrize("name", ["msrvtt", "ivqa"])
def test_videoqa_train_gpu(golden, name):
    from frozenbilm_amd.optim import FusedAdam

    g = golden("G10_videoqa", raw=True)
    cfg, P, m = hip_model(N_ANS, 10, g["a2tok"], train=True)
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    before = {k: m.get_param(k).detach().clone() for k in DELTA_KEYS}
    opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.95))
    batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 3, 4, seed=102, dataset_name=name)
    stats = P_vqa.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device(DEV), 0, name, args, max_norm=0.1)
    ref = _j(g, f"train_{name}_stats")
    for k in ref:
        assert abs(stats[k] - ref[k]) < 2e-2, (k, stats[k], ref[k])
    for k in DELTA_KEYS:  # three clipped Adam steps: same direction as the reference's fp32 run
        d = (m.get_param(k).detach() - before[k]).cpu()
        assert cosine(d, torch.as_tensor(g[f"train_{name}_delta/{k}"])) > 0.9, k


def test_mc_evaluate_and_train_gpu(golden):
    from frozenbilm_amd.optim import FusedAdam

    g = golden("G11_mc", raw=True)
    cfg, P, m = hip_model(2, 11, g["a2tok"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 4, 3, 4, seed=111)
    with torch.no_grad():
        sc = P_mc.candidate_scores(m, tok, batches[0], torch.device(DEV), args).cpu()
    ref_sc = torch.as_tensor(g["eval_scores0"])
    assert (sc - ref_sc).abs().max().item() < 2e-2
    results, acc = P_mc.evaluate(m, tok, ListLoader(batches, mc=4), torch.device(DEV), "how2qa", args)
    ref_results = _j(g, "eval_results")
    flips = sum(results[q]["pred"] != ref_results[q]["pred"] for q in results)
    assert flips <= 1 and abs(acc - float(g["eval_acc"][0])) <= 1.0 / len(results) + 1e-9
    # predictions agree wherever the reference separates best and second-best candidate by more than 2e-2
    for i, q in enumerate(batches[0]["qid"]):
        top2 = ref_sc[i].topk(2).values
        if (top2[0] - top2[1]).item() > 2e-2:
            assert results[q]["pred"] == ref_results[q]["pred"]
    m.train()
    before = {k: m.get_param(k).detach().clone() for k in DELTA_KEYS}
    opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.95))
    tb = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 4, 3, 4, seed=113)
    stats = P_mc.train_one_epoch(m, tok, ListLoader(tb, mc=4), opt, torch.device(DEV), 0, args, max_norm=0.1)
    ref = _j(g, "train_stats")
    for k in ref:
        assert abs(stats[k] - ref[k]) < 2e-2, (k, stats[k], ref[k])
    for k in DELTA_KEYS:
        d = (m.get_param(k).detach() - before[k]).cpu()
        assert cosine(d, torch.as_tensor(g[f"train_delta/{k}"])) > 0.9, k


def test_main_train_and_evaluate_loops_gpu():
    """main.py:24-153 loops end to end: same seeded host-side mask_tokens draws on both sides, HIP model vs oracle model."""
    from frozenbilm_amd import main as P_main
    from frozenbilm_amd.optim import FusedAdam
    from tests.downstream_fixtures import make_videotext_batches

    from frozenbilm_amd.model.config import DebertaV2Config
    from frozenbilm_amd.model.deberta import DebertaV2ForMaskedLM

    cfg = tiny(0)
    P = O.synth_params(cfg, seed=21, std=0.08, ln_jitter=0.1)
    c = DebertaV2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        max_position_embeddings=cfg.max_position_embeddings, position_buckets=cfg.position_buckets,
                        layer_norm_eps=cfg.layer_norm_eps, conv_kernel_size=cfg.conv_kernel_size,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = DebertaV2ForMaskedLM(c, max_feats=cfg.max_feats, features_dim=cfg.features_dim, ds_factor_attn=cfg.ds_factor_attn,
                             ds_factor_ff=cfg.ds_factor_ff, dropout=0.0)
    m.load_state_dict(P, strict=False)
    m.to(DEV)
    om = OracleModel(tiny(0), P)
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=31)
    torch.manual_seed(7)
    ref = P_main.evaluate(om, tok, ListLoader(batches), torch.device("cpu"), args)
    torch.manual_seed(7)
    got = P_main.evaluate(m, tok, ListLoader(batches), torch.device(DEV), args)
    assert got.keys() == ref.keys()
    for k in ref:
        assert abs(got[k] - ref[k]) < 2e-2, (k, got[k], ref[k])
    torch.manual_seed(8)
    oopt = torch.optim.Adam([p for p in om.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    ref = P_main.train_one_epoch(om, tok, ListLoader(batches), oopt, torch.device("cpu"), 0, args, 0.1)
    torch.manual_seed(8)
    got = P_main.train_one_epoch(m, tok, ListLoader(batches), FusedAdam(m, lr=1e-3, betas=(0.9, 0.95)), torch.device(DEV), 0, args, 0.1)
    for k in ref:
        assert abs(got[k] - ref[k]) < 2e-2, (k, got[k], ref[k])
    # device-side corruption kernel instead of the host sampler: runs, finite, loss in the same range
    args.device_mask_tokens = True
    got2 = P_main.evaluate(m, tok, ListLoader(batches), torch.device(DEV), args)
    assert abs(got2["loss"] - got["loss"]) < 1.0 and got2["loss"] > 0


def test_mask_row_head_and_batched_candidates_match_the_reference_shaped_path(golden):
    """SURVEY 8(f) at speed: (1) the prediction head run on the [MASK] rows only (``logit_rows``) returns exactly the rows
    the reference-shaped path selects from the full logits; (2) mc: all candidates of a batch in ONE forward of C.B samples
    give the scores -- and bit-identical predicted ids -- of the reference's one-forward-per-candidate loop."""
    g = golden("G11_mc", raw=True)
    cfg, P, m = hip_model(2, 11, g["a2tok"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_choices=4, n_batches=2, B=5, seed=311)
    from frozenbilm_amd.loops import tokenize, video_inputs

    with torch.no_grad():
        for b in batches:
            video, vmask = video_inputs(b, torch.device(DEV))
            enc = tokenize(tok, b["text"][1], args)
            feed = dict(video=video, video_mask=vmask, input_ids=enc["input_ids"].to(DEV), attention_mask=enc["attention_mask"].to(DEV))
            full = P_vqa.mask_row_logits(m(**feed)["logits"], enc["input_ids"], tok, args)
            rows = P_vqa.answer_logits(m, tok, enc["input_ids"], args, **feed)
            assert rows.shape == full.shape and torch.equal(rows, full)
            batched = P_mc.candidate_scores(m, tok, b, torch.device(DEV), args)
            args.mc_sequential = True
            seq = P_mc.candidate_scores(m, tok, b, torch.device(DEV), args)
            args.mc_sequential = False
            assert batched.shape == seq.shape == (5, 4)
            assert (batched - seq).abs().max().item() < 2e-3  # a sample's logits do not depend on its batch (padding masked)
            assert torch.equal(batched.max(1).indices, seq.max(1).indices)
    # and the whole loop still reproduces the reference's results on its own golden batches
    ref_batches = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, n_choices=4, n_batches=3, B=4, seed=111)
    results, acc = P_mc.evaluate(m, tok, ListLoader(ref_batches, mc=4), torch.device(DEV), "how2qa", args)
    ref = _j(g, "eval_results")
    flips = sum(int(results[k]["pred"] != ref[str(k)]["pred"]) for k in results)
    assert flips <= 1, flips


# ------------------------------------------------------------------------------------------------ BASELINE configs 4 / 5 at FULL size
def _full_size_setup(n_ans, B, Lt, seed):
    """DeBERTa-v2-XLarge (24 layers) + adapters with an answer head, and a synthetic batch of BASELINE's shape: T = 10 x 1024
    features, ragged texts of up to Lt tokens with exactly one [MASK] each (what bench.py --workload videoqa / mc feeds)."""
    from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM

    cfg = DebertaV2Config()
    torch.manual_seed(seed)
    m = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1, n_ans=n_ans)
    m.to(DEV).eval()
    g = torch.Generator().manual_seed(seed + 1)
    a2tok = torch.randint(5, cfg.vocab_size, (n_ans, 5), generator=g)
    a2tok = a2tok * (torch.arange(5)[None] < torch.randint(1, 6, (n_ans, 1), generator=g))
    m.set_answer_embeddings(a2tok.to(DEV))
    MASK = 128000

    class Tok:
        mask_token_id, pad_token_id, sep_token_id = MASK, 0, 2

        def __call__(self, text, **kw):
            ids = torch.stack(text)
            return {"input_ids": ids, "attention_mask": (ids != 0).long()}

    def texts(s):
        gg = torch.Generator().manual_seed(s)
        tlen = torch.randint(Lt // 8, Lt + 1, (B,), generator=gg)
        tlen[-1] = Lt
        ids = torch.randint(5, 127000, (B, Lt), generator=gg) * (torch.arange(Lt)[None] < tlen[:, None])
        ids[torch.arange(B), torch.stack([torch.randint(1, int(t), (1,), generator=gg) for t in tlen]).view(-1)] = MASK
        return list(ids)

    video = torch.randn(B, 10, 1024, generator=g).half().float()
    vlen = torch.randint(1, 11, (B,), generator=g)
    import types

    args = types.SimpleNamespace(max_feats=10, use_video=True, suffix="", use_context=True, max_tokens=Lt, print_freq=10 ** 9)
    return m, Tok(), texts, video, vlen, args


def _sub(batch, sl):
    out = {}
    for k, v in batch.items():
        if k == "text" and isinstance(v[0], list):
            out[k] = [c[sl] for c in v]
        else:
            out[k] = v[sl]
    return out


@pytest.mark.slow
def test_full_size_videoqa_config4_batch_independence():
    """BASELINE config 4 at its full size (24 layers, B = 32, L = 256, n_ans = 1000, head on the [MASK] rows): the
    product's `videoqa.evaluate` on the whole batch returns, question for question, what it returns on the two halves --
    answer logits within 1e-3 and the same top-10 answer ids wherever neighbouring candidates are separated by more."""
    B = 32
    m, tok, texts, video, vlen, args = _full_size_setup(1000, B, 256, seed=21)
    batch = dict(video=video, video_len=vlen, qid=list(range(B)), type=[0] * B, text=texts(11),
                 answer_id=torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3)))

    def logits_of(bd):
        from frozenbilm_amd.loops import tokenize, video_inputs

        v, vm = video_inputs(bd, torch.device(DEV))
        enc = tokenize(tok, bd["text"], args)
        with torch.no_grad():
            return P_vqa.answer_logits(m, tok, enc["input_ids"], args, video=v, video_mask=vm, input_ids=enc["input_ids"].to(DEV),
                                       attention_mask=enc["attention_mask"].to(DEV)).float()

    full = logits_of(batch)
    halves = torch.cat([logits_of(_sub(batch, slice(0, 16))), logits_of(_sub(batch, slice(16, 32)))], 0)
    assert full.shape == (B, 1000) and torch.isfinite(full).all()
    err = (full - halves).abs().max().item()
    print(f"config 4 full size: max |logit(B=32) - logit(2 x B=16)| = {err:.2e}")
    assert err < 1e-3, err
    top_f, top_h = full.topk(11, -1), halves.topk(11, -1)
    gaps = (top_f.values[:, :-1] - top_f.values[:, 1:]).min(-1).values  # smallest separation among the first 11 candidates
    clear = gaps > 4 * err + 1e-6
    assert clear.float().mean().item() > 0.5
    assert torch.equal(top_f.indices[clear][:, :10], top_h.indices[clear][:, :10])  # answer indices bit-exact
    # and through the loop itself: same per-question predictions and metrics
    res_f, out_f = P_vqa.evaluate(m, tok, ListLoader([batch]), torch.device(DEV), "msrvtt", args, thresholds=[1, 10])
    res_h, out_h = P_vqa.evaluate(m, tok, ListLoader([_sub(batch, slice(0, 16)), _sub(batch, slice(16, 32))]), torch.device(DEV),
                                  "msrvtt", args, thresholds=[1, 10])
    for q in range(B):
        if bool(clear[q]):
            assert res_f[q]["pred"] == res_h[q]["pred"], q
    assert abs(out_f["acc1"] - out_h["acc1"]) <= (~clear).sum().item() / B + 1e-9


@pytest.mark.slow
def test_full_size_mc_config5_candidates_in_one_forward():
    """BASELINE config 5 at its full size (24 layers, B = 8, 4 candidates, S = 512): the scores of `mc.candidate_scores`
    -- all 32 candidate rows in ONE forward -- equal the reference's one-forward-per-candidate loop (mc.py:140-165,
    `mc_sequential`) within 1e-3, and the chosen candidate is the same wherever the margin is larger."""
    B, C = 8, 4
    m, tok, texts, video, vlen, args = _full_size_setup(2, B, 502, seed=31)
    batch = dict(video=video, video_len=vlen, qid=list(range(B)), type=[0] * B, text=[texts(11 + c) for c in range(C)],
                 answer_id=torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(4)))
    with torch.no_grad():
        one = P_mc.candidate_scores(m, tok, batch, torch.device(DEV), args).float()
        args.mc_sequential = True
        seq = P_mc.candidate_scores(m, tok, batch, torch.device(DEV), args).float()
    assert one.shape == (B, C) and torch.isfinite(one).all()
    err = (one - seq).abs().max().item()
    print(f"config 5 full size: max |score(one forward) - score(per candidate)| = {err:.2e}")
    assert err < 1e-3, err
    s2 = seq.topk(2, -1).values
    clear = (s2[:, 0] - s2[:, 1]) > 4 * err + 1e-6
    assert torch.equal(one.argmax(-1)[clear], seq.argmax(-1)[clear])


def test_inference_graph_replay_matches_the_eager_forward(golden):
    """`model.inference_graphs = True`: the [MASK]-row inference forward of the evaluate loops runs as one hipGraph replay
    (text padded to the length bucket, inputs copied into static buffers).  Replays reproduce the eager logits to the
    batch-independence tolerance (the padded length changes tile shapes), give IDENTICAL results when the same batch is
    replayed, follow a parameter update made between two replays, and a second batch shape gets its own graph."""
    g = golden("G10_videoqa", raw=True)
    cfg, P, m = hip_model(N_ANS, 10, g["a2tok"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 3, 4, seed=101, dataset_name="msrvtt")

    def logits_of(bd):
        from frozenbilm_amd.loops import tokenize, video_inputs

        v, vm = video_inputs(bd, torch.device(DEV))
        enc = tokenize(tok, bd["text"], args)
        with torch.no_grad():
            return P_vqa.answer_logits(m, tok, enc["input_ids"], args, video=v, video_mask=vm, input_ids=enc["input_ids"].to(DEV),
                                       attention_mask=enc["attention_mask"].to(DEV)).float()

    eager = [logits_of(b) for b in batches]
    m.inference_graphs = True
    first = [logits_of(b) for b in batches]
    assert m.inference_graphs, "capture failed"
    again = [logits_of(b) for b in batches]
    for e, a, b in zip(eager, first, again):
        assert a.shape == e.shape and (a - e).abs().max().item() < 2e-3, (a - e).abs().max().item()
        assert torch.equal(a, b)  # a replay is deterministic
    assert len(m.__dict__["_graph_cache"]) >= 1
    # a parameter update between replays is picked up (operands are rebuilt in place, outside the graph)
    with torch.no_grad():
        p = m.get_param("deberta.encoder.layer.1.output.adapter.up.bias")
        p.add_(torch.linspace(-0.3, 0.3, p.numel(), device=p.device))  # (a uniform shift would vanish in the LayerNorm)
    m.inference_graphs = False
    moved_eager = logits_of(batches[0])
    m.inference_graphs = True
    moved_graph = logits_of(batches[0])
    assert (moved_eager - eager[0]).abs().max().item() > 1e-3
    assert (moved_graph - moved_eager).abs().max().item() < 2e-3
    # the evaluate loop itself with the switch on returns the same predictions
    args.inference_graphs = True
    res_g, out_g = P_vqa.evaluate(m, tok, ListLoader(batches), torch.device(DEV), "msrvtt", args, thresholds=[1, 10])
    m.inference_graphs = False
    args.inference_graphs = False
    res_e, out_e = P_vqa.evaluate(m, tok, ListLoader(batches), torch.device(DEV), "msrvtt", args, thresholds=[1, 10])
    assert out_g == out_e and all(res_g[q]["pred"][0] == res_e[q]["pred"][0] for q in res_e)
