"""SURVEY 8(f) ranks 1-2 -- the downstream loops (videoqa.py, mc.py: BASELINE configs 4 and 5).

The goldens G10 / G11 hold what the REFERENCE's own loops returned on seeded synthetic batches (results dicts, metrics,
losses, parameter updates).  On CPU the product's host loops are driven with the oracle model (pins the loop logic and
the oracle together, predicted ids bit-exact); on the GPU the same loops run the HIP model.
"""
import json

import pytest
import torch

from frozenbilm_amd import mc as P_mc
from frozenbilm_amd import videoqa as P_vqa
from oracle import deberta_oracle as O
from oracle.model_wrapper import OracleModel
from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_mc_batches, make_videoqa_batches
from tests.golden.make_goldens import _tiny_cfg

N_ANS = 40
DELTA_KEYS = ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
              "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight")


def tiny(n_ans):
    return _tiny_cfg(n_ans=n_ans, max_feats=4, vocab_size=300, max_position_embeddings=128)


def oracle_model(n_ans, seed, a2tok):
    cfg = tiny(n_ans)
    return cfg, OracleModel(cfg, O.synth_params(cfg, seed=seed, std=0.08, ln_jitter=0.1), a2tok)


def _j(g, key):
    return json.loads(str(g[key]))


def cosine(a, b):
    return (a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm() + 1e-30)).item()


@torch.no_grad()
def mask_probs(m, tok, batches, args, dev="cpu"):
    """answer probabilities at the [MASK] row per qid (the quantity the loops rank)"""
    from frozenbilm_amd.util.misc import get_mask

    out = {}
    for b in batches:
        enc = tok(b["text"])
        ids, att = enc["input_ids"].clone().to(dev), enc["attention_mask"].clone().to(dev)
        if not args.suffix and not args.use_context:
            att[ids == tok.sep_token_id] = 0
            ids[ids == tok.sep_token_id] = tok.pad_token_id
        lg = m(video=b["video"].to(dev), video_mask=get_mask(b["video_len"], b["video"].size(1)).to(dev), input_ids=ids,
               attention_mask=att)["logits"]
        pr = P_vqa.mask_row_logits(lg, enc["input_ids"], tok, args).float().softmax(-1).cpu()
        for q, p in zip(b["qid"], pr):
            out[q] = p
    return out


def same_ranking(pred, ref_pred, probs, tol):
    """ids equal rank by rank; a swap is only accepted between answers whose probabilities differ by < tol (relative):
    the reference's own fp32 ordering of such a pair is decided by rounding noise."""
    for a, b in zip(pred, ref_pred):
        if a != b and abs(probs[a] - probs[b]) > tol * max(probs[a], probs[b]):
            return False
    return sorted(pred) == sorted(ref_pred) or all(
        abs(probs[a] - probs[b]) <= tol * max(probs[a], probs[b]) for a, b in zip(pred, ref_pred) if a != b)


# ------------------------------------------------------------------------------------------ CPU: host loops x oracle
@pytest.mark.parametrize("name", ["msrvtt", "ivqa"])
def test_videoqa_evaluate_matches_reference(golden, name):
    g = golden("G10_videoqa", raw=True)
    cfg, m = oracle_model(N_ANS, 10, torch.as_tensor(g["a2tok"]))
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 3, 4, seed=101, dataset_name=name)
    results, metrics = P_vqa.evaluate(m, tok, ListLoader(batches), torch.device("cpu"), name, args, thresholds=[1, 10],
                                      split="test", type_map={0: "a", 1: "b"})
    ref_results, ref_metrics = _j(g, f"eval_{name}_results"), _j(g, f"eval_{name}_metrics")
    assert results.keys() == ref_results.keys()
    probs = mask_probs(m, tok, batches, args)
    exact = 0
    for q in results:
        # top-10 answer ids: bit-exact except between answers tied to 1e-5 relative in fp32
        assert same_ranking(results[q]["pred"], ref_results[q]["pred"], probs[q], 1e-5), (q, results[q]["pred"], ref_results[q]["pred"])
        exact += results[q]["pred"] == ref_results[q]["pred"]
        assert results[q]["pred"][0] == ref_results[q]["pred"][0]
        assert results[q]["gt"] == ref_results[q]["gt"] and results[q]["type"] == ref_results[q]["type"]
        assert results[q]["acc1"] == ref_results[q]["acc1"] and results[q]["acc10"] == ref_results[q]["acc10"]
    assert exact >= len(results) - 2
    assert {k: float(v) for k, v in metrics.items()} == pytest.approx(ref_metrics, abs=1e-9)
    assert (probs["q0"] - torch.as_tensor(g[f"eval_{name}_probs0"])[0]).abs().max().item() < 1e-6


@pytest.mark.parametrize("name", ["msrvtt", "ivqa"])
def test_videoqa_train_matches_reference(golden, name):
    g = golden("G10_videoqa", raw=True)
    cfg, m = oracle_model(N_ANS, 10, torch.as_tensor(g["a2tok"]))
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    before = {k: v.detach().clone() for k, v in m.named_ref_parameters().items()}
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    batches = make_videoqa_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, N_ANS, 3, 4, seed=102, dataset_name=name)
    stats = P_vqa.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device("cpu"), 0, name, args, max_norm=0.1)
    ref = _j(g, f"train_{name}_stats")
    assert stats.keys() == ref.keys()
    for k in ref:
        assert abs(stats[k] - ref[k]) < 2e-5, (k, stats[k], ref[k])
    after = m.named_ref_parameters()
    for k in DELTA_KEYS:
        d = after[k].detach() - before[k]
        assert cosine(d, torch.as_tensor(g[f"train_{name}_delta/{k}"])) > 0.999, k


def test_mc_evaluate_and_train_match_reference(golden):
    g = golden("G11_mc", raw=True)
    cfg, m = oracle_model(2, 11, torch.as_tensor(g["a2tok"]))
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 4, 3, 4, seed=111)
    results, acc = P_mc.evaluate(m, tok, ListLoader(batches, mc=4), torch.device("cpu"), "how2qa", args)
    assert results == _j(g, "eval_results")
    assert acc == pytest.approx(float(g["eval_acc"][0]))
    hidden = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 4, 1, 4, seed=112, with_gt=False)
    results, acc = P_mc.evaluate(m, tok, ListLoader(hidden, mc=4), torch.device("cpu"), "how2qa", args)
    assert results == _j(g, "eval_hidden_results") and acc == 0
    with torch.no_grad():
        sc = P_mc.candidate_scores(m, tok, batches[0], torch.device("cpu"), args)
    assert (sc - torch.as_tensor(g["eval_scores0"])).abs().max().item() < 1e-5
    # training: balanced BCE through 4 forwards per step
    before = {k: v.detach().clone() for k, v in m.named_ref_parameters().items()}
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    tb = make_mc_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 4, 3, 4, seed=113)
    stats = P_mc.train_one_epoch(m, tok, ListLoader(tb, mc=4), opt, torch.device("cpu"), 0, args, max_norm=0.1)
    ref = _j(g, "train_stats")
    for k in ref:
        assert abs(stats[k] - ref[k]) < 2e-5, (k, stats[k], ref[k])
    after = m.named_ref_parameters()
    for k in DELTA_KEYS:
        d = after[k].detach() - before[k]
        assert cosine(d, torch.as_tensor(g[f"train_delta/{k}"])) > 0.999, k


def test_main_loops_match_reference(golden):
    """main.py:24-153: the product's train_one_epoch / evaluate (host loop, host mask_tokens with the reference's RNG
    consumption order) driven with the oracle model vs what the reference's own loops returned."""
    from frozenbilm_amd import main as P_main
    from tests.downstream_fixtures import make_videotext_batches

    g = golden("G13_main_loops", raw=True)
    cfg = _tiny_cfg(max_feats=4, vocab_size=300, max_position_embeddings=128)
    m = OracleModel(cfg, O.synth_params(cfg, seed=21, std=0.08, ln_jitter=0.1))
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=31)
    torch.manual_seed(7)
    ev = P_main.evaluate(m, tok, ListLoader(batches), torch.device("cpu"), args)
    ref = _j(g, "eval_stats")
    assert ev.keys() == ref.keys()
    for k in ref:
        assert abs(ev[k] - ref[k]) < 2e-5, (k, ev[k], ref[k])
    torch.manual_seed(8)
    before = {k: v.detach().clone() for k, v in m.named_ref_parameters().items()}
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    tr = P_main.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device("cpu"), 0, args, 0.1)
    ref = _j(g, "train_stats")
    assert tr.keys() == ref.keys()
    for k in ref:
        assert abs(tr[k] - ref[k]) < 2e-5, (k, tr[k], ref[k])
    after = m.named_ref_parameters()
    for k in DELTA_KEYS:
        assert cosine(after[k].detach() - before[k], torch.as_tensor(g[f"train_delta/{k}"])) > 0.999, k
