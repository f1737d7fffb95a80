"""SURVEY 8(f) ranks 3-4 on CPU: the input-side dataset (bit-exact against the reference's dataset class, golden G12)
and checkpoints in the reference's schema."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import deberta_oracle as O
from tests.golden.make_goldens import _tiny_cfg, write_feature_fixture


def test_videotext_dataset_matches_reference(golden, tmp_path):
    from frozenbilm_amd.datasets import VideoText_Dataset, videotext_collate_fn

    g = golden("G12_dataset", raw=True)
    csv_path, feats = write_feature_fixture(str(tmp_path))
    ds = VideoText_Dataset(csv_path, feats, max_feats=10, features_dim=16)
    assert len(ds) == 9
    batch = videotext_collate_fn([ds[i] for i in range(len(ds))])
    assert batch["video"].dtype == torch.float32 and batch["video_len"].dtype == torch.long
    assert torch.equal(batch["video"], torch.from_numpy(g["video"]))  # fp16 -> fp32 reads, index picks: bit-exact
    assert torch.equal(batch["video_len"], torch.from_numpy(g["video_len"]))
    assert batch["text"] == [str(t) for t in g["text"]]
    assert batch["video_len"].tolist() == [3, 10, 10, 10, 1, 10, 10, 0, 0]  # short, exact, long, ..., missing, corrupt


def test_subsample_indices_formula():
    from frozenbilm_amd.datasets.videotext_dataset import subsample_indices

    for n in (11, 25, 47, 1000):
        assert subsample_indices(n, 10).tolist() == [(j * n) // 10 for j in range(10)]


def _model(cfg, P):
    from frozenbilm_amd.model.config import DebertaV2Config
    from frozenbilm_amd.model.deberta import DebertaV2ForMaskedLM

    c = DebertaV2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        max_position_embeddings=cfg.max_position_embeddings, position_buckets=cfg.position_buckets,
                        layer_norm_eps=cfg.layer_norm_eps, conv_kernel_size=cfg.conv_kernel_size)
    m = DebertaV2ForMaskedLM(c, max_feats=cfg.max_feats, features_dim=cfg.features_dim, ds_factor_attn=cfg.ds_factor_attn,
                             ds_factor_ff=cfg.ds_factor_ff, n_ans=cfg.n_ans)
    if P is not None:
        m.load_state_dict(P, strict=False)
    return m


def test_checkpoint_reference_schema_roundtrip(tmp_path):
    from frozenbilm_amd.util.checkpoint import load_checkpoint, save_checkpoint

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1)
    m = _model(cfg, P)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    args = argparse.Namespace(lr=1e-3, epochs=2)
    path = os.path.join(tmp_path, "checkpoint0000.pth")
    save_checkpoint(m, opt, 0, args, path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "args", "fbl"}  # "fbl": dropout-stream position (extra key)
    # state_dict keys are the reference's, including its position_ids buffer
    assert set(ck["model"]) == set(O.param_shapes(cfg)) | {"deberta.embeddings.position_ids"}
    m2 = _model(cfg, None)
    opt2 = torch.optim.Adam([p for p in m2.parameters() if p.requires_grad], lr=5e-4)
    _, start = load_checkpoint(m2, path, opt2, resume=True)
    assert start == 1 and opt2.param_groups[0]["lr"] == 1e-3
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # a reference-produced file also carries buffers / tied weights the build does not own: ignored, not an error
    ck["model"]["lm_predictions.lm_head.decoder.weight"] = torch.zeros(cfg.vocab_size, cfg.hidden_size)
    torch.save(ck, path)
    load_checkpoint(_model(cfg, None), path)
    # adapter-only file (the trained subset) on top of other frozen weights
    path2 = os.path.join(tmp_path, "adapters.pth")
    save_checkpoint(m, None, 3, args, path2, trainable_only=True)
    sub = torch.load(path2, map_location="cpu", weights_only=False)["model"]
    assert all(O.is_trainable(k) for k in sub) and len(sub) == sum(O.is_trainable(k) for k in O.param_shapes(cfg))
    P2 = O.synth_params(cfg, seed=4, std=0.05)
    m3 = _model(cfg, P2)
    load_checkpoint(m3, path2)
    sd3 = m3.state_dict()
    for k in O.param_shapes(cfg):
        want = P[k] if O.is_trainable(k) else P2[k]
        assert torch.equal(sd3[k], want), k
    # shape mismatch is an error, not a silent skip
    sub["deberta.embeddings.linear_video.weight"] = torch.zeros(3, 3)
    torch.save({"model": sub, "optimizer": None, "epoch": 0, "args": args}, path2)
    with pytest.raises(RuntimeError):
        load_checkpoint(m3, path2)


REF_CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G16_reference_checkpoint.pth")
DELTA_KEYS = ("deberta.embeddings.linear_video.weight", "deberta.encoder.layer.1.output.adapter.up.weight",
              "deberta.encoder.layer.0.attention.output.adapter.down.weight", "deberta.encoder.LayerNorm.weight")


def test_reference_written_checkpoint_loads_and_resumes(golden):
    """SURVEY 8(f) rank 4 against a file the REFERENCE wrote (golden G16: main.py's epoch + its own save_on_master call,
    main.py:290-300 / util/dist.py:195-198): the product's load_checkpoint restores model and optimizer, and the product's
    loops continue exactly where the reference's continue (evaluate, then one resumed epoch)."""
    import json

    from frozenbilm_amd import main as P_main
    from frozenbilm_amd.util.checkpoint import load_checkpoint
    from oracle.model_wrapper import OracleModel
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_videotext_batches

    meta = golden("G16_checkpoint_meta", raw=True)
    order = [str(n) for n in meta["trainable_order"]]
    cfg = _tiny_cfg(max_feats=4, vocab_size=300, max_position_embeddings=128)
    m = _model(cfg, O.synth_params(cfg, seed=23, std=0.08))  # different weights: everything must come from the file
    assert [n for n, p in m.named_parameters() if p.requires_grad] == order, "optimizer-state index space differs"
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, betas=(0.9, 0.95))
    ck, start = load_checkpoint(m, REF_CKPT, opt, resume=True)
    assert start == 1 and set(ck) == {"model", "optimizer", "epoch", "args"}
    sd = m.state_dict()
    for k, v in ck["model"].items():
        if "lm_head.decoder" in k:
            continue  # the reference's untied copy of the word embeddings (not a parameter of the path)
        assert torch.equal(sd[k], v), k
    assert opt.param_groups[0]["lr"] == 1e-3
    for i, p in enumerate(opt.param_groups[0]["params"]):
        assert torch.equal(opt.state[p]["exp_avg"], ck["optimizer"]["state"][i]["exp_avg"]), order[i]
    # continue on the CPU oracle model with the loaded weights, driven by the PRODUCT's loops
    P = {k: v.clone() for k, v in sd.items() if "position_ids" not in k}
    om = OracleModel(cfg, P)
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=32)
    torch.manual_seed(10)
    ev = P_main.evaluate(om, tok, ListLoader(batches), torch.device("cpu"), args)
    ref = json.loads(str(meta["eval_stats"]))
    for k in ref:
        assert abs(ev[k] - ref[k]) < 2e-5, (k, ev[k], ref[k])
    named = om.named_ref_parameters()
    opt2 = torch.optim.Adam([named[n] for n in order], lr=5e-4, betas=(0.9, 0.95))
    opt2.load_state_dict(ck["optimizer"])
    before = {k: named[k].detach().clone() for k in DELTA_KEYS}
    torch.manual_seed(11)
    tr = P_main.train_one_epoch(om, tok, ListLoader(batches), opt2, torch.device("cpu"), 1, args, 0.1)
    ref = json.loads(str(meta["resume_train_stats"]))
    for k in ref:
        assert abs(tr[k] - ref[k]) < 2e-5, (k, tr[k], ref[k])
    for k in DELTA_KEYS:
        d = (om.named_ref_parameters()[k].detach() - before[k]).flatten()
        r = torch.as_tensor(meta[f"resume_delta/{k}"]).flatten()
        assert torch.dot(d, r) / (d.norm() * r.norm() + 1e-30) > 0.999, k
        assert (d - r).norm() / r.norm() < 2e-3, k
