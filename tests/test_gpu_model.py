"""Model-level parity on a real MI355X: the HIP pipeline (through the reference-shaped Python boundary) against the
golden vectors captured from the reference and against the CPU oracle on seeded inputs.

Tolerances (bf16 MFMA operands, fp32 accumulation / residual stream): logits max-abs <= 5e-2 (BASELINE north_star),
loss <= 2e-2 abs, trainable gradients <= 6% of each tensor's max (bf16 operand rounding through a 3..24-layer chain),
integer outputs (top-k ids on well separated logits) exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import deberta_oracle as O  # noqa: E402
from tests.golden.make_goldens import _tiny_cfg, synth_batch  # noqa: E402
from tests.gpu_refs import stats  # noqa: E402

DEV = "cuda"


def build(cfg: O.OracleConfig, P, train=False, engine_options=None, ft_ln=True):
    from frozenbilm_amd.model.config import DebertaV2Config
    from frozenbilm_amd.model.deberta import DebertaV2ForMaskedLM

    c = DebertaV2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        max_position_embeddings=cfg.max_position_embeddings, position_buckets=cfg.position_buckets,
                        layer_norm_eps=cfg.layer_norm_eps, conv_kernel_size=cfg.conv_kernel_size)
    m = DebertaV2ForMaskedLM(c, max_feats=cfg.max_feats, features_dim=cfg.features_dim, ds_factor_attn=cfg.ds_factor_attn,
                             ds_factor_ff=cfg.ds_factor_ff, n_ans=cfg.n_ans, ft_ln=ft_ln)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k for k in missing), missing
    if engine_options:
        m.engine_options = dict(engine_options)
    m.to(DEV)
    m.train(train)
    return m


def to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def test_tiny_forward_golden(golden):
    g = golden("G5_tiny_model")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    with torch.no_grad():
        out = m(**to_dev(batch), output_hidden_states=True)
    hs = torch.stack([h.float().cpu() for h in out.hidden_states], 0)
    print(stats("hidden_states", hs, g["hidden_states"]))
    print(stats("logits", out.logits.cpu(), g["logits"]))
    err_h = (hs - g["hidden_states"]).abs().max().item()
    assert err_h < 5e-2, err_h
    assert (out.logits.cpu() - g["logits"]).abs().max().item() < 5e-2
    assert abs(out.loss.item() - g["loss"].item()) < 2e-2
    assert out["loss"] is out.loss  # item + attribute access like MaskedLMOutput


def test_output_attentions_golden(golden):
    """output_attentions=True (model/deberta.py:1414-1427): one [B, heads, S, S] probability tensor per encoder layer, equal
    to the reference's `attentions` tuple (G5c); masked pairs and masked query rows exactly 0; the other outputs of the
    call are those of the plain forward; a training-mode request (attention dropout live) is refused."""
    g5, gc = golden("G5_tiny_model"), golden("G5c_tiny_attentions")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    batch = to_dev({k[3:]: v for k, v in g5.items() if k.startswith("in.")})
    with torch.no_grad():
        out = m(**batch, output_attentions=True)
        plain = m(**batch)
    assert plain.attentions is None and torch.equal(out.logits, plain.logits)
    att = out.attentions
    ref = gc["attentions"]
    assert isinstance(att, tuple) and len(att) == cfg.num_hidden_layers == ref.shape[0]
    worst = 0.0
    for li, a in enumerate(att):
        assert a.shape == ref[li].shape and a.dtype == torch.float32
        worst = max(worst, (a.cpu() - ref[li]).abs().max().item())
        assert (a.cpu()[ref[li] == 0] == 0).all(), "masked pairs / masked query rows must be exactly 0"
        rows = a.sum(-1).cpu()
        live = ref[li].sum(-1) > 0.5
        assert (rows[live] - 1).abs().max().item() < 2e-2
    print(f"attention probabilities vs reference: max-abs err {worst:.2e}")
    assert worst < 2e-2, worst
    assert (out.logits.float().cpu() - g5["logits"]).abs().max().item() < 5e-2
    tup = m(**batch, output_attentions=True, return_dict=False)
    assert isinstance(tup[-1], tuple) and len(tup[-1]) == cfg.num_hidden_layers
    m.train()
    with pytest.raises(NotImplementedError):
        m(**batch, output_attentions=True)


def test_tiny_text_only_golden(golden):
    g5, gb = golden("G5_tiny_model"), golden("G5b_tiny_textonly")
    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1))
    with torch.no_grad():
        out = m(input_ids=g5["in.input_ids"].to(DEV), attention_mask=g5["in.attention_mask"].to(DEV))
    assert out.logits.shape == gb["logits"].shape
    assert (out.logits.cpu() - gb["logits"]).abs().max().item() < 5e-2


def test_tiny_backward_golden(golden):
    g = golden("G5_tiny_model")
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=5, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)  # eval mode: dropout off, gradients on (the reference goldens were taken the same way)
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    out = m(**to_dev(batch))
    out.loss.backward()
    bad = []
    n = 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        ref = g["grad." + name]
        n += 1
        got = p.grad.float().cpu()
        # bf16 vs fp32 flips a fraction f ~ 0.5 % of the adapters' ReLU gates (pre-activations near 0).  The gradient of
        # adapter.down.{weight,bias} is a random-signed sum over rows, so its relative error is ~sqrt(2 f) ~ 10 % for ANY
        # number of rows: those tensors get a looser bound here and an exact, gate-matched check in
        # test_gpu_kernels.py::test_adapter_module_gate_matched (reference fed the same bf16 operands).
        fro = (got - ref).norm().item() / max(ref.norm().item(), 1e-9)
        rel = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        lim = 0.25 if "adapter.down" in name else 6e-2
        if fro > lim or rel > 0.4:
            bad.append((name, round(fro, 4), round(rel, 4)))
    assert n == len([k for k in g if k.startswith("grad.")])
    assert not bad, bad
    # a second backward accumulates (p.grad += ...), zero_grad(set_to_none) resets
    g1 = {nm: p.grad.clone() for nm, p in m.named_parameters() if p.requires_grad}
    out2 = m(**to_dev(batch))
    out2.loss.backward()
    for nm, p in m.named_parameters():
        if p.requires_grad:
            assert torch.allclose(p.grad, 2 * g1[nm], rtol=1e-3, atol=1e-6), nm
    m.zero_grad(set_to_none=True)
    m(**to_dev(batch)).loss.backward()
    for nm, p in m.named_parameters():
        if p.requires_grad:
            assert torch.allclose(p.grad, g1[nm], rtol=1e-3, atol=1e-6), nm


def test_backward_vs_oracle_larger_batch():
    """N = 8 x 130 rows: gate-flip noise averages out (~1/sqrt(N)); a systematic backward bug would not."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=41, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    batch = synth_batch(cfg, B=8, L=120, seed=7)
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    ref = O.forward(P, cfg, **batch)
    ref["loss"].backward()
    out = m(**to_dev(batch))
    out.loss.backward()
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2
    worst = []
    for name, p in m.named_parameters():
        if p.requires_grad:
            r = P[name].grad
            fro = (p.grad.float().cpu() - r).norm().item() / max(r.norm().item(), 1e-9)
            worst.append((round(fro, 4), name))
    worst.sort(reverse=True)
    print("worst relative Frobenius grad errors:", worst[:8])
    strict = [w for w in worst if "adapter.down" not in w[1]]
    assert strict[0][0] < 4e-2, strict[:6]
    assert worst[0][0] < 0.18, worst[:6]  # ReLU-gate flips, see test_tiny_backward_golden (measured 12.5 %; the rest < 4 %)


def test_answer_head_golden(golden):
    g = golden("G9_answers")
    cfg = _tiny_cfg(n_ans=50)
    P = O.synth_params(cfg, seed=9, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    m.set_answer_embeddings(g["a2tok"].to(DEV))
    assert torch.allclose(m.get_param("answer_embeddings.weight").cpu(), g["answer_embeddings"], atol=1e-6)
    batch = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    with torch.no_grad():
        out = m(**to_dev(batch))
    lg = out.logits.float().cpu()
    assert lg.shape == g["logits"].shape
    assert (lg - g["logits"]).abs().max().item() < 5e-2
    # top-1 answer ids: exact wherever the reference's top-2 margin exceeds the tolerance
    ref_sorted = g["logits"].sort(-1, descending=True).values
    clear = (ref_sorted[..., 0] - ref_sorted[..., 1]) > 0.1
    assert torch.equal(lg.argmax(-1)[clear], g["top10"][..., 0][clear])


def test_ragged_lengths_vs_oracle():
    """S not a multiple of 16/64, empty video rows, padded text: bf16 path vs the fp32 CPU oracle."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=21, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    for (B, Lt, seed) in ((1, 1, 1), (2, 60, 2), (5, 131, 3)):
        batch = synth_batch(cfg, B=B, L=Lt, seed=seed) if Lt > 2 else dict(
            video=torch.randn(1, 10, cfg.features_dim), video_mask=torch.zeros(1, 10, dtype=torch.long),
            input_ids=torch.tensor([[7]]), attention_mask=torch.ones(1, 1, dtype=torch.long), labels=torch.tensor([[7]]))
        with torch.no_grad():
            ref = O.forward(P, cfg, **batch)
            out = m(**to_dev(batch))
        err = (out.logits.cpu() - ref["logits"]).abs().max().item()
        assert err < 5e-2, (B, Lt, err)
        assert abs(out.loss.item() - ref["loss"].item()) < 2e-2


@pytest.mark.parametrize("B,Lt", [(1, 2), (3, 54), (2, 55), (2, 63), (3, 64), (1, 118), (2, 119), (4, 17), (7, 33)])
def test_shape_sweep_forward_backward_vs_oracle(B, Lt):
    """Sequence lengths on and around the 64-row tile boundaries of the attention kernels (S = 10 + Lt = 64, 65, 73, 74,
    128, 129), odd batch sizes, a two-token text: logits, loss and the gradients that do not pass an adapter ReLU gate
    against the fp32 CPU oracle (gate-flip noise of the adapter.down tensors: test_tiny_backward_golden)."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=33, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    batch = synth_batch(cfg, B=B, L=Lt, seed=100 + B * 7 + Lt)
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    ref = O.forward(P, cfg, **batch)
    ref["loss"].backward()
    out = m(**to_dev(batch))
    out.loss.backward()
    assert (out.logits.float().cpu() - ref["logits"].detach()).abs().max().item() < 5e-2
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2
    bad = []
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        r = P[name].grad
        fro = (p.grad.float().cpu() - r).norm().item() / max(r.norm().item(), 1e-9)
        lim = 0.30 if "adapter.down" in name else 8e-2  # (few rows: a single flipped gate weighs more than at N = 1000)
        if fro > lim:
            bad.append((name, round(fro, 4)))
    assert not bad, bad


def test_sequence_limit_error():
    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=1, std=0.05))
    ids = torch.ones(1, 503, dtype=torch.long, device=DEV)
    with pytest.raises(RuntimeError):
        m(input_ids=ids, attention_mask=torch.ones_like(ids), video=torch.zeros(1, 10, cfg.features_dim, device=DEV))
    with pytest.raises(ValueError):
        m()


def test_train_mode_dropout_and_step():
    """train(): counter-based dropout is live (outputs differ from eval, differ step to step), loss finite, an
    optimizer step changes only trainable parameters and lowers the loss on a repeated batch."""
    from frozenbilm_amd.optim import FusedAdam

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=31, std=0.05, ln_jitter=0.1)
    m = build(cfg, P, train=True)
    batch = to_dev(synth_batch(cfg, B=4, L=40, seed=5))
    frozen_before = m.get_param("deberta.encoder.layer.0.intermediate.dense.weight").clone()
    opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.95))
    l0 = m(**batch).loss
    l1 = m(**batch).loss
    assert torch.isfinite(l0) and l0.item() != l1.item()
    m.eval()
    with torch.no_grad():
        le0 = m(**batch).loss.item()
    m.train()
    for _ in range(8):
        opt.zero_grad()
        loss = m(**batch).loss
        loss.backward()
        opt.step(clip_max_norm=1.0)
    m.eval()
    with torch.no_grad():
        le1 = m(**batch).loss.item()
    assert le1 < le0, (le0, le1)
    assert torch.equal(frozen_before, m.get_param("deberta.encoder.layer.0.intermediate.dense.weight"))


@pytest.mark.slow
def test_xlarge_golden(golden):
    """True DeBERTa-v2-XLarge dims, seeded weights (regenerated bit-identically on this box), B=2, S=266."""
    g = golden("G6_xlarge")
    cfg = O.OracleConfig()
    P = O.synth_params(cfg, seed=0)
    m = build(cfg, P)
    del P
    batch = synth_batch(cfg, B=2, L=256, seed=66)
    with torch.no_grad():
        out = m(**to_dev(batch))
    lg = out.logits
    sl = lg[:, ::19, ::997].float().cpu()
    print(stats("xlarge logits slice", sl, g["logits_slice"]))
    err = (sl - g["logits_slice"]).abs()
    row0 = (lg[0, 12, :2048].float().cpu() - g["logits_row0"]).abs()
    print("row0 max err", row0.max().item(), "loss", out.loss.item(), "ref", g["loss"].item())
    # north_star: logits within 5e-2 (bf16); SURVEY section 7 notes a naive mixed-precision run only meets it at p99.9
    assert err.max().item() < 5e-2 and row0.max().item() < 5e-2
    assert abs(out.loss.item() - g["loss"].item()) < 2e-2
    agree = (lg.argmax(-1).cpu() == g["argmax"]).float().mean().item()
    print("argmax agreement", agree)
    assert agree > 0.99  # measured 0.998 - 1.000 (rounds 3 / 4)


def test_max_length_s512_forward_backward_vs_oracle():
    """BASELINE config 5 regime: S = T + L = 512 exactly (L = 502 with ASR context), so every log bucket of the relative-
    position map and all 8 key tiles are exercised; 2-way answer head; forward + gradients vs the fp32 oracle."""
    cfg = _tiny_cfg(n_ans=2, num_hidden_layers=2)
    P = O.synth_params(cfg, seed=55, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    g = torch.Generator().manual_seed(12)
    a2tok = torch.randint(5, cfg.vocab_size, (2, 3), generator=g)
    m.set_answer_embeddings(a2tok.to(DEV))
    Po = {k: v.clone() for k, v in P.items()}
    Po["answer_embeddings.weight"] = O.answer_embeddings(a2tok, P, cfg)
    batch = synth_batch(cfg, B=3, L=502, seed=13)
    batch["attention_mask"][0] = 1  # one sample at the full length
    batch["input_ids"][0] = torch.randint(5, cfg.vocab_size, (502,), generator=g)
    batch.pop("labels")
    for k, v in Po.items():
        v.requires_grad_(O.is_trainable(k))
    ref = O.forward(Po, cfg, **batch)
    out = m(**to_dev(batch))
    assert out.logits.shape == (3, 512, 2)
    valid = torch.cat([batch["video_mask"], batch["attention_mask"]], 1).bool()
    err = (out.logits.float().cpu() - ref["logits"])[valid].abs().max().item()
    assert err < 5e-2, err
    # mc.py-style score on one row per sample, back-propagated through the logits
    rows = torch.tensor([300, 40, 17])
    pick = lambda lg: lg[torch.arange(3), rows].softmax(-1)[:, 0]
    tgt = torch.tensor([1.0, 0.0, 1.0])
    torch.nn.functional.binary_cross_entropy(pick(ref["logits"]), tgt).backward()
    torch.nn.functional.binary_cross_entropy(pick(out.logits), tgt.to(DEV)).backward()
    bad = []
    for name, p in m.named_parameters():
        if p.requires_grad:
            r = Po[name].grad
            fro = (p.grad.float().cpu() - r).norm().item() / max(r.norm().item(), 1e-12)
            if fro > (0.25 if "adapter.down" in name else 6e-2):
                bad.append((name, round(fro, 4)))
    assert not bad, bad[:8]


def test_step_state_dies_with_the_loss_no_gc_needed():
    """A step's saved activations must be freed by reference counting the moment its loss is dropped: with the cyclic
    GC switched off, no Run object may outlive its step (a cycle here once grew the allocator by ~5 GB per step)."""
    import gc

    from frozenbilm_amd.engine import Run
    from frozenbilm_amd.optim import FusedAdam

    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=61, std=0.05, ln_jitter=0.1), train=True)
    opt = FusedAdam(m, lr=1e-4)
    batch = to_dev(synth_batch(cfg, B=3, L=30, seed=8))
    gc.collect()
    gc.disable()
    try:
        for _ in range(3):
            opt.zero_grad(set_to_none=False)
            loss = m(**batch).loss
            loss.backward()
            opt.step(clip_max_norm=0.1)
        del loss
        torch.cuda.synchronize()
        live = [o for o in gc.get_objects() if isinstance(o, Run)]
        assert not live, f"{len(live)} Run objects kept alive by reference cycles"
        out = m(**batch)  # an output that is still referenced keeps exactly its own state
        assert len([o for o in gc.get_objects() if isinstance(o, Run)]) == 1
        del out
        assert not [o for o in gc.get_objects() if isinstance(o, Run)]
    finally:
        gc.enable()


def test_full_size_batch_independence_xlarge():
    """BASELINE config 2 at its full size (DeBERTa-v2-XLarge dims, 24 layers, B=32, T=10x1024, L=256, ragged): a sample's
    logits must not depend on what else is in the batch -- the B=32 forward equals the two B=16 halves row for row
    (padding, key-tile skipping, tile quantisation and the round-filling GEMM splits all change with the batch)."""
    import bench as Bn
    from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM

    cfg = DebertaV2Config()
    torch.manual_seed(0)
    m = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1).to(DEV).eval()
    batch = Bn.synth_batch(32, 10, 1024, 256, cfg.vocab_size, seed=3, device=torch.device(DEV))
    cols = torch.arange(0, cfg.vocab_size, 997, device=DEV)

    def run(sl):
        with torch.no_grad():
            out = m(**{k: v[sl] for k, v in batch.items()})
        lg = out.logits
        return lg[:, :, cols].float().clone(), lg.argmax(-1), out.loss.item()

    full, am_full, loss_full = run(slice(0, 32))
    lo, am_lo, loss_lo = run(slice(0, 16))
    hi, am_hi, loss_hi = run(slice(16, 32))
    valid = torch.cat([batch["video_mask"], batch["attention_mask"]], 1).bool()
    halves, am_halves = torch.cat([lo, hi], 0), torch.cat([am_lo, am_hi], 0)
    err = (full - halves)[valid].abs().max().item()
    assert err < 1e-3, err
    assert torch.equal(am_full[valid], am_halves[valid])  # token indices bit-exact
    n = (batch["labels"] != -100).sum(1).float()
    want = (loss_lo * n[:16].sum() + loss_hi * n[16:].sum()) / n.sum()  # mean over labelled rows
    assert abs(loss_full - want.item()) < 1e-4, (loss_full, want.item())


def test_full_size_backward_linearity_xlarge():
    """Full BASELINE size, backward: gradients are linear in the loss scale (2 x loss -> 2 x every gradient; power-of-two
    scaling commutes with bf16 rounding, so only the fp32 atomic orders differ), finite, and non-zero for all 298
    trainable tensors."""
    import bench as Bn
    from frozenbilm_amd.model import DebertaV2Config, DebertaV2ForMaskedLM

    cfg = DebertaV2Config()
    torch.manual_seed(0)
    m = DebertaV2ForMaskedLM(cfg, max_feats=10, features_dim=1024, ds_factor_attn=8, ds_factor_ff=8, dropout=0.1).to(DEV).eval()
    batch = Bn.synth_batch(32, 10, 1024, 256, cfg.vocab_size, seed=4, device=torch.device(DEV))
    grads = []
    for scale in (1.0, 2.0):
        for p in m.parameters():
            p.grad = None
        loss = m(**batch).loss
        (loss * scale).backward()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad})
    assert len(grads[0]) == 298
    worst = 0.0
    for n, g1 in grads[0].items():
        g2 = grads[1][n]
        assert torch.isfinite(g1).all() and g1.abs().max().item() > 0, n
        rel = (g2 - 2 * g1).norm().item() / (2 * g1.norm().item() + 1e-30)
        worst = max(worst, rel)
    assert worst < 2e-3, worst


# ------------------------------------------------------------------------------------------------ round-2 parity pins
def _rel_fro(a, b):
    return (a - b).norm().item() / max(b.norm().item(), 1e-12)


def _tight_cfgs():
    xl = O.OracleConfig()  # true xlarge dimensions (H=1536, 24 heads, I=6144, adapters 192, F=1024) ...
    xl.num_hidden_layers = 4  # ... with a reduced layer count and vocabulary so the CPU oracle finishes in seconds
    xl.vocab_size = 4096
    return [pytest.param(_tiny_cfg(), 8, 120, id="tiny"), pytest.param(xl, 2, 96, id="xlarge-dims-4-layers", marks=pytest.mark.slow)]


@pytest.mark.parametrize("cfg,B,Lt", _tight_cfgs())
def test_gradients_vs_bf16_operand_oracle_tight(cfg, B, Lt):
    """Every trainable gradient against the oracle run with the HIP path's arithmetic contract (matrix-multiply operands
    rounded to bf16, fp32 accumulation: oracle.bf16_operands) AND the ReLU gates the GPU run took (oracle.adapter_gates,
    read from the saved bottleneck activations).  That removes the one discontinuity through which rounding noise turns
    into a 10 % difference of d(adapter.down): every gradient, adapter.down included, must then agree to a few per cent
    -- a 10 % systematic error in dW_down, which the 25 % bound against the pure-fp32 reference cannot see, fails here.
    Runs at the tiny configuration and at the true xlarge dimensions (H = 1536, 64-wide heads, I = 6144, 192-wide
    bottlenecks, merged dense + adapter-down GEMMs on the 8-phase tiles) with 4 layers."""
    tiny = cfg.hidden_size < 1024
    P = O.synth_params(cfg, seed=41, std=0.05 if tiny else 0.02, ln_jitter=0.1)
    m = build(cfg, P)
    batch = synth_batch(cfg, B=B, L=Lt, seed=7)
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    out = m(**to_dev(batch))
    # the ReLU gates the GPU run actually took (saved bottleneck activations z > 0), in execution order
    gates = []
    for sv in out.__dict__["_run"].layers:
        gates += [(sv.z1[:, : cfg.hidden_size // cfg.ds_factor_attn] > 0).cpu(), (sv.z2[:, : cfg.hidden_size // cfg.ds_factor_ff] > 0).cpu()]
    out.loss.backward()
    with O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]):
        ref = O.forward(P, cfg, **batch)
        ref["loss"].backward()
    assert abs(out.loss.item() - ref["loss"].item()) < 5e-3
    assert (out.logits.float().cpu() - ref["logits"]).abs().max().item() < 3e-2
    worst = sorted(((round(_rel_fro(p.grad.float().cpu(), P[n].grad), 4), n) for n, p in m.named_parameters() if p.requires_grad),
                   reverse=True)
    print("worst relative Frobenius grad errors vs the bf16-operand oracle:", worst[:8])
    assert worst[0][0] < 2e-2, worst[:8]  # measured: 0.7 % worst, adapter.down included


# ------------------------------------------------------------------------------------------------ round-5: train-mode parity
def _gates_of(run, cfg):
    g = []
    for sv in run.layers:
        if cfg.ds_factor_attn:
            g.append((sv.z1[:, : cfg.hidden_size // cfg.ds_factor_attn] > 0).cpu())
        if cfg.ds_factor_ff:
            g.append((sv.z2[:, : cfg.hidden_size // cfg.ds_factor_ff] > 0).cpu())
    return g


@pytest.mark.parametrize("cfg,B,Lt", _tight_cfgs())
def test_train_mode_parity_with_replayed_dropout_masks(cfg, B, Lt):
    """The configuration the headline is timed in -- train(), dropout live at every reference site (model/deberta.py:142-240,
    258, 332, 403, 779, 796, 1054; model/adapter.py:40-41; main.py:34) -- against the oracle.  The HIP kernels draw their masks
    from a counter-based hash of (seed, element index); tests/dropout_replay.py restates that hash on the host, rebuilds the
    masks of THIS forward from the per-site seeds the engine recorded and hands them to the oracle's dropout sites (whose
    place / scale / order is pinned against the reference itself by golden G17).  Loss, logits and EVERY trainable gradient
    must then agree like the eval-mode checks do: a dropout site at the wrong place, a wrong 1/(1-p) in one of the fused
    epilogues (merged GEMM bottleneck, adapter tail, ln_fwd, attention kernels) or a site missing in backward fails here.
    Checked for the eager step and for a `training_graphs` replay (per-site constants + device seed word)."""
    from tests.dropout_replay import ReplayedMasks

    tiny = cfg.hidden_size < 1024
    P = O.synth_params(cfg, seed=43, std=0.05 if tiny else 0.02, ln_jitter=0.1)
    batch = synth_batch(cfg, B=B, L=Lt, seed=9)
    results = {}
    for graphs in (False, True):
        torch.manual_seed(777)  # (the model's mask stream mixes torch.initial_seed(): both models draw the same masks)
        m = build(cfg, P, train=True)
        m.training_graphs = graphs
        out = m(**to_dev(batch))
        run = out.__dict__["_run"]
        if not graphs:
            c = m.config
            masks = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob,
                                  m.adapter_dropout)
            gates = _gates_of(run, cfg)
            assert len(run.layers) == cfg.num_hidden_layers + 1 and masks.seed_emb != 0
        else:
            assert len(m.__dict__.get("_train_graphs", {})) == 1, "the step was not served by a captured graph"
        out.loss.backward()
        logits = out.logits.float().cpu()
        results[graphs] = (out.loss.item(), logits, {n: p.grad.float().cpu().clone() for n, p in m.named_parameters() if p.requires_grad})
        del m, out, run
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    with O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]), O.dropout_masks(masks):
        ref = O.forward(P, cfg, **batch)
        ref["loss"].backward()
    assert masks.exhausted(), masks.asked
    with torch.no_grad():
        ev = O.forward(P, cfg, **batch)["loss"].item()
    assert abs(ev - ref["loss"].item()) > 1e-3, "dropout made no difference: the comparison would prove nothing"
    for graphs, (loss, logits, grads) in results.items():
        tag = "graph replay" if graphs else "eager"
        dl = abs(loss - ref["loss"].item())
        de = (logits - ref["logits"]).abs().max().item()
        worst = sorted(((round(_rel_fro(g_, P[n].grad), 4), n) for n, g_ in grads.items()), reverse=True)
        print(f"train-mode parity [{tag}]: loss {loss:.5f} vs {ref['loss'].item():.5f} (eval-mode oracle {ev:.5f}), "
              f"logits max-abs {de:.3e}, worst grads {worst[:4]}")
        assert dl < 2e-2, (tag, dl)
        assert de < 5e-2, (tag, de)
        assert worst[0][0] < 2e-2, (tag, worst[:8])
    # eager and replayed steps drew the same masks: identical results
    assert results[False][0] == results[True][0]
    for n in results[False][2]:
        assert torch.equal(results[False][2][n], results[True][2][n]), n


@pytest.mark.parametrize("seed", [1000, 8919, 16838, 103947])
def test_train_mode_parity_other_mask_streams(seed):
    """The train-mode comparison must not depend on which masks a run happens to draw: four more mask streams (torch seeds) of the
    tiny model, eager step, gate-matched bf16-operand oracle with replayed masks -- measured over 24 streams: worst gradient
    <= 1.2 % (max-abs relative), where the pure-fp32 oracle sees 1-38 % on d(adapter.down) from ReLU gate flips alone."""
    from tests.dropout_replay import ReplayedMasks

    cfg = _tiny_cfg()
    B, Lt = 4, 40
    P = O.synth_params(cfg, seed=43, std=0.05, ln_jitter=0.1)
    batch = synth_batch(cfg, B=B, L=Lt, seed=9)
    torch.manual_seed(seed)
    m = build(cfg, P, train=True)
    out = m(**to_dev(batch))
    run = out.__dict__["_run"]
    c = m.config
    masks = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob, m.adapter_dropout)
    gates = _gates_of(run, cfg)
    out.loss.backward()
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
        v.grad = None
    with O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]), O.dropout_masks(masks):
        ref = O.forward(P, cfg, **batch)
        ref["loss"].backward()
    assert masks.exhausted()
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2
    assert (out.logits.float().cpu() - ref["logits"]).abs().max().item() < 5e-2
    worst = max((_rel_fro(p.grad.float().cpu(), P[n].grad), n) for n, p in m.named_parameters() if p.requires_grad)
    assert worst[0] < 2e-2, worst


def test_training_trajectory_follows_the_oracle_over_optimizer_steps():
    """Six steps of the reference's update rule (main.py:67-84: forward in train mode, backward, clip_grad_norm_(0.1), Adam with
    lr / betas of args.py) on six different batches, the HIP path (FusedAdam: clip folded into the flat update) beside the oracle
    (torch.optim.Adam + clip_grad_norm_ on ITS OWN parameter trajectory).  Every step the oracle sees the dropout masks and ReLU
    gates of the HIP step; nothing else is shared, so an error in one step's gradients, in the clip factor or in the moments
    compounds into the next step's loss.  Checked: loss of every step, worst gradient of every step, and the direction of the
    accumulated parameter update of every trainable tensor."""
    from frozenbilm_amd.optim import FusedAdam
    from tests.dropout_replay import ReplayedMasks

    cfg = _tiny_cfg()
    B, Lt, steps, lr, clip = 4, 40, 6, 3e-4, 0.1
    P0 = O.synth_params(cfg, seed=43, std=0.05, ln_jitter=0.1)
    torch.manual_seed(4242)
    m = build(cfg, P0, train=True)
    opt = FusedAdam(m, lr=lr, betas=(0.9, 0.95))
    Pr = {k: v.clone() for k, v in P0.items()}
    for k, v in Pr.items():
        v.requires_grad_(O.is_trainable(k))
    train_names = [k for k in Pr if O.is_trainable(k)]
    opt_ref = torch.optim.Adam([Pr[k] for k in train_names], lr=lr, betas=(0.9, 0.95))
    c = m.config
    log = []
    for s in range(steps):
        batch = synth_batch(cfg, B=B, L=Lt, seed=50 + s)
        opt.zero_grad()
        out = m(**to_dev(batch))
        run = out.__dict__["_run"]
        masks = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob, m.adapter_dropout)
        gates = _gates_of(run, cfg)
        out.loss.backward()
        grads = {n: p.grad.float().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}
        opt.step(clip_max_norm=clip)
        gn = opt.grad_norm().item()
        opt_ref.zero_grad()
        with O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]), O.dropout_masks(masks):
            ref = O.forward(Pr, cfg, **batch)
            ref["loss"].backward()
        assert masks.exhausted()
        worst = max((_rel_fro(grads[n], Pr[n].grad), n) for n in train_names)
        gn_ref = torch.nn.utils.clip_grad_norm_([Pr[k] for k in train_names], clip).item()
        opt_ref.step()
        log.append((out.loss.item(), ref["loss"].item(), worst, gn, gn_ref))
        del out, run
    for s, (lh, lo, worst, gn, gn_ref) in enumerate(log):
        print(f"step {s}: loss {lh:.5f} vs {lo:.5f}, worst grad {worst[0]:.4f} ({worst[1]}), grad norm {gn:.4f} vs {gn_ref:.4f}")
    assert abs(log[0][0] - log[-1][0]) > 1e-3  # (different batches, moving parameters: not one number six times)
    for lh, lo, worst, gn, gn_ref in log:
        assert abs(lh - lo) < 2e-2, log
        assert worst[0] < 3e-2, log
        assert abs(gn - gn_ref) < 2e-2 * gn_ref, log
    # accumulated update of every trainable tensor: same direction, same size.  Adam's first steps move every element by ~lr
    # whatever the size of its gradient, so the elements whose gradient is smaller than the two paths' arithmetic difference
    # move in a random direction in both: the bound is on the direction of the whole tensor, not per element.
    cos, size = [], []
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        dh = (p.detach().float().cpu() - P0[n]).flatten()
        dr = (Pr[n].detach() - P0[n]).flatten()
        assert dr.abs().max().item() > 0.5 * lr, n  # the parameter moved
        cos.append((torch.nn.functional.cosine_similarity(dh, dr, dim=0).item(), n))
        size.append((dh.norm() / dr.norm()).item())
    print("update cosine, worst five:", sorted(cos)[:5], "size ratio range:", min(size), max(size))
    assert min(cos)[0] > 0.9, sorted(cos)[:5]
    assert 0.9 < min(size) and max(size) < 1.1, (min(size), max(size))
    frozen = "deberta.encoder.layer.0.intermediate.dense.weight"
    assert torch.equal(m.get_param(frozen).float().cpu(), P0[frozen])


@pytest.mark.parametrize("ds_attn,ds_ff,ft_ln", [(0, 0, True), (8, 8, False), (0, 8, True), (8, 0, False)],
                         ids=["no-adapters", "ft_ln-off", "ffn-adapter-only", "attn-adapter-only+ft_ln-off"])
def test_freeze_policy_flag_variants_vs_oracle(ds_attn, ds_ff, ft_ln):
    """The constructor flags of the reference's ablations on the GPU: `ds_factor_attn / ds_factor_ff = 0` (no adapter at that
    site, model/deberta.py:252,326) and `ft_ln=False` (LayerNorms frozen, :1152-1158; args.py:333-337).  Trainable set = the
    reference's substring rule; logits / loss / every trainable gradient against the oracle (bf16-operand mode, the GPU's ReLU
    gates), frozen parameters receive no gradient, train mode included (dropout masks replayed)."""
    from tests.dropout_replay import ReplayedMasks

    cfg = _tiny_cfg(ds_factor_attn=ds_attn, ds_factor_ff=ds_ff)
    P = O.synth_params(cfg, seed=61, std=0.05, ln_jitter=0.1)
    B, Lt = 4, 50
    batch = synth_batch(cfg, B=B, L=Lt, seed=13)
    for train in (False, True):
        torch.manual_seed(99)
        m = build(cfg, P, train=train, ft_ln=ft_ln)
        want_train = sorted(k for k in P if O.is_trainable(k, ft_ln=ft_ln))
        assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == want_train
        assert want_train, "nothing trainable"
        out = m(**to_dev(batch))
        run = out.__dict__["_run"]
        gates = _gates_of(run, cfg)
        masks = None
        if train:
            c = m.config
            masks = ReplayedMasks(run, cfg, cfg.num_attention_heads, c.hidden_dropout_prob, c.attention_probs_dropout_prob,
                                  m.adapter_dropout)
        out.loss.backward()
        for k, v in P.items():
            v.requires_grad_(O.is_trainable(k, ft_ln=ft_ln))
            v.grad = None
        import contextlib

        with O.bf16_operands(), O.adapter_gates([g_.view(B, -1, g_.shape[-1]) for g_ in gates]), \
                (O.dropout_masks(masks) if train else contextlib.nullcontext()):
            ref = O.forward(P, cfg, **batch)
            ref["loss"].backward()
        if train:
            assert masks.exhausted(), masks.asked
        assert abs(out.loss.item() - ref["loss"].item()) < 2e-2
        assert (out.logits.float().cpu() - ref["logits"]).abs().max().item() < 5e-2
        worst = sorted(((round(_rel_fro(p.grad.float().cpu(), P[n].grad), 4), n) for n, p in m.named_parameters() if p.requires_grad),
                       reverse=True)
        print(f"flags ds_attn={ds_attn} ds_ff={ds_ff} ft_ln={ft_ln} train={train}: worst grads {worst[:3]}")
        assert worst[0][0] < 2e-2, worst[:6]
        assert all(p.grad is None for n, p in m.named_parameters() if not p.requires_grad)
        del m, out, run


@pytest.mark.slow
def test_xlarge_backward_golden(golden):
    """G6b: the reference's own backward at true xlarge dims (24 layers, H=1536, B=2, S=266): norms of all 298 trainable
    gradients and the stored full vectors / strided slices (linear_video, the stand-alone LayerNorms, everything trainable
    in layers 0, 12, 23, conv LayerNorm)."""
    from tests.golden.make_goldens import xl_grad_slices

    g = golden("G6b_xlarge_backward", raw=True)
    cfg = O.OracleConfig()
    P = O.synth_params(cfg, seed=0)
    m = build(cfg, P)
    del P
    batch = synth_batch(cfg, B=2, L=256, seed=67)
    out = m(**to_dev(batch))
    out.loss.backward()
    assert abs(out.loss.item() - float(g["loss"])) < 2e-2
    sl = out.logits[:, ::19, ::997].float().cpu()
    assert (sl - torch.from_numpy(g["logits_slice"])).abs().max().item() < 5e-2
    names = [str(n) for n in g["names"]]
    got = {n: p.grad.float().cpu() for n, p in m.named_parameters() if p.requires_grad}
    assert sorted(names) == sorted(got)
    norm_err = sorted(((abs(got[n].double().norm().item() - r) / max(r, 1e-30), n) for n, r in zip(names, g["norms"])), reverse=True)
    print("worst gradient-norm errors:", [(round(e, 4), n) for e, n in norm_err[:6]])
    assert norm_err[0][0] < 0.03, norm_err[:6]  # measured: 1.8 - 2.05 % worst (an adapter.down tensor), < 1 % for the rest
    bad, worst = [], []
    for k in g:
        if not k.startswith("grad."):
            continue
        n = k[5:]
        ref = torch.from_numpy(g[k])
        mine = xl_grad_slices(n, got[n])
        fro = _rel_fro(mine, ref)
        worst.append((round(fro, 4), n))
        # adapter.down: ReLU-gate flips against the pure-fp32 reference (measured <= 12.5 %; the gate-matched test above
        # holds the same tensors to 2 %); everything else measured <= 3.7 %
        if fro > (0.17 if "adapter.down" in n else 5e-2):  # measured 12.2 - 12.9 % / 4.0 %
            bad.append((n, round(fro, 4)))
    worst.sort(reverse=True)
    print("worst relative Frobenius errors on the stored slices:", worst[:8])
    assert not bad, bad[:8]


@pytest.mark.slow
def test_xlarge_downstream_goldens(golden):
    """G15 (cfg 4: n_ans = 1000 answer head, one [MASK] row per sample, top-10 ids) and G14 (cfg 5: S = 512, Yes/No head)
    at true xlarge dims, against the reference's outputs."""
    g15, g14 = golden("G15_xlarge_videoqa"), golden("G14_xlarge_mc")
    cfg = O.OracleConfig(n_ans=1000)
    P = O.synth_params(cfg, seed=0)
    m = build(cfg, P)
    del P
    gen = torch.Generator().manual_seed(150)
    a2tok = torch.randint(5, cfg.vocab_size, (1000, 5), generator=gen)
    alen = torch.randint(1, 6, (1000,), generator=gen)
    a2tok = a2tok * (torch.arange(5)[None] < alen[:, None])
    m.set_answer_embeddings(a2tok.to(DEV))
    MASK = int(g15["mask_id"])
    batch = synth_batch(cfg, B=2, L=256, seed=151)
    batch.pop("labels")
    batch["input_ids"][torch.arange(2), g15["mask_pos"]] = MASK
    with torch.no_grad():
        lg = m(**to_dev(batch)).logits.float().cpu()
    rows = lg[:, cfg.max_feats:][batch["input_ids"] == MASK]
    assert (rows - g15["mask_logits"]).abs().max().item() < 5e-2
    assert (lg[:, ::19, ::97] - g15["logits_slice"]).abs().max().item() < 5e-2
    probs = rows.softmax(-1)
    assert (probs - g15["mask_probs"]).abs().max().item() < 2e-3
    # token indices: the top-10 answer ids agree at every rank the reference separates from both neighbours by more than
    # twice the logit tolerance, and the top-10 sets overlap in all but near-tied entries
    ref_l = g15["mask_logits"].sort(-1, descending=True).values
    top = probs.topk(10, -1).indices
    checked = 0
    for b in range(2):
        for r in range(10):
            if (r == 0 or ref_l[b, r - 1] - ref_l[b, r] > 0.1) and ref_l[b, r] - ref_l[b, r + 1] > 0.1:
                assert top[b, r] == g15["top10"][b, r], (b, r)
                checked += 1
        assert len(set(top[b].tolist()) & set(g15["top10"][b].tolist())) >= 8
    print("top-10 ranks checked exactly:", checked)
    # ---- cfg 5 regime on the same weights: 2-answer head, S = 512
    m.set_answer_embeddings(g14["a2tok"].to(DEV))
    b5 = synth_batch(cfg, B=2, L=502, seed=141)
    b5.pop("labels")
    b5["input_ids"][torch.arange(2), g14["mask_pos"]] = MASK
    with torch.no_grad():
        lg5 = m(**to_dev(b5)).logits.float().cpu()
    assert lg5.shape == (2, 512, 2)
    valid = torch.cat([b5["video_mask"], b5["attention_mask"]], 1).bool()
    assert (lg5 - g14["logits"])[valid].abs().max().item() < 5e-2
    score = lg5[:, cfg.max_feats:][b5["input_ids"] == MASK].softmax(-1)[:, 0]
    assert (score - g14["score"]).abs().max().item() < 1e-2


def test_lazy_logits_every_accessor_and_autograd():
    """With labels the forward computes the labelled rows only; the [B,S,V] logits are filled on first access through ANY
    accessor of the output mapping, equal to an eager run, and stay wired into autograd (loss + a loss on the logits)."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=77, std=0.05, ln_jitter=0.1)
    m = build(cfg, P)
    batch = to_dev(synth_batch(cfg, B=3, L=33, seed=9))
    nolab = {k: v for k, v in batch.items() if k != "labels"}
    with torch.no_grad():
        want = m(**nolab).logits.clone()  # eager path (no labels)
    for access in (lambda o: o.logits, lambda o: o["logits"], lambda o: o[1], lambda o: o.get("logits"),
                   lambda o: dict(o)["logits"], lambda o: list(o.values())[1], lambda o: dict(o.items())["logits"],
                   lambda o: {**o}["logits"]):
        with torch.no_grad():
            out = m(**batch)
            assert out.__dict__.get("_fill") is not None  # nothing filled yet
            got = access(out)
        assert torch.equal(got, want)
    # differentiable: d(loss + f(logits)) == d loss + d f(logits) (two separate runs)
    def grads(use_loss, use_logits):
        for p in m.parameters():
            p.grad = None
        out = m(**batch)
        tot = 0
        if use_loss:
            tot = tot + out.loss
        if use_logits:
            tot = tot + out.logits[:, :, :7].float().pow(2).mean()
        tot.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}

    ga, gb, gab = grads(True, False), grads(False, True), grads(True, True)
    for n in ga:  # (three separate runs: bf16 rounding noise of the backward operands differs between them)
        assert _rel_fro(gab[n], ga[n] + gb[n]) < 2e-2, (n, _rel_fro(gab[n], ga[n] + gb[n]))
    assert max(v.abs().max().item() for v in gb.values()) > 0


def test_fused_adam_resumes_reference_checkpoint(golden):
    """A checkpoint the reference wrote (G16: torch.optim.Adam state, main.py:290-300) restores the HIP model and the fused
    optimizer; evaluate and one resumed epoch through the product's loops track what the reference did with the same file."""
    import json
    import os

    from frozenbilm_amd import main as P_main
    from frozenbilm_amd.optim import FusedAdam
    from frozenbilm_amd.util.checkpoint import load_checkpoint, save_checkpoint
    from tests.downstream_fixtures import Args, ListLoader, StubTokenizer, make_videotext_batches
    from tests.test_inputs_checkpoint import DELTA_KEYS, REF_CKPT

    meta = golden("G16_checkpoint_meta", raw=True)
    cfg = _tiny_cfg(max_feats=4, vocab_size=300, max_position_embeddings=128)
    m = build(cfg, O.synth_params(cfg, seed=23, std=0.08))
    m.adapter_dropout = 0.0
    m.config.hidden_dropout_prob = m.config.attention_probs_dropout_prob = 0.0  # G16 was taken with dropout 0
    opt = FusedAdam(m, lr=5e-4, betas=(0.9, 0.95))
    ck, start = load_checkpoint(m, REF_CKPT, opt, resume=True)
    assert start == 1 and opt._step == 3 and opt.param_groups[0]["lr"] == 1e-3
    eng = m.engine()
    n0 = "deberta.encoder.layer.1.output.adapter.up.weight"
    i0 = [n for n, p in m.named_parameters() if p.requires_grad].index(n0)
    o, k = eng.offsets[n0], eng.named[n0].numel()
    assert torch.equal(opt._m[o:o + k].cpu().view_as(ck["optimizer"]["state"][i0]["exp_avg"]), ck["optimizer"]["state"][i0]["exp_avg"])
    tok, args = StubTokenizer(cfg.vocab_size), Args(max_feats=cfg.max_feats)
    batches = make_videotext_batches(cfg.vocab_size, cfg.max_feats, cfg.features_dim, 3, 6, seed=32)
    torch.manual_seed(10)
    ev = P_main.evaluate(m, tok, ListLoader(batches), torch.device(DEV), args)
    ref = json.loads(str(meta["eval_stats"]))
    assert abs(ev["loss"] - ref["loss"]) < 2e-2, (ev, ref)
    before = {k_: m.get_param(k_).detach().clone() for k_ in DELTA_KEYS}
    torch.manual_seed(11)
    tr = P_main.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device(DEV), 1, args, 0.1)
    ref = json.loads(str(meta["resume_train_stats"]))
    assert abs(tr["loss"] - ref["loss"]) < 2e-2, (tr, ref)
    for k_ in DELTA_KEYS:
        d = (m.get_param(k_).detach() - before[k_]).flatten().float().cpu()
        r = torch.as_tensor(meta[f"resume_delta/{k_}"]).flatten()
        cos = (torch.dot(d, r) / (d.norm() * r.norm() + 1e-30)).item()
        assert cos > 0.9, (k_, cos)  # Adam's sign-like update amplifies bf16 gradient noise on near-zero entries
    # and a file written HERE carries torch.optim.Adam's schema (the reference can resume it)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "c.pth")
        save_checkpoint(m, opt, 1, args, path)
        ck2 = torch.load(path, map_location="cpu", weights_only=False)
        assert set(ck2["optimizer"]) == {"state", "param_groups"} and len(ck2["optimizer"]["state"]) == 46
        named = [p for p in m.parameters() if p.requires_grad]
        topt = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in named], lr=1.0)
        topt.load_state_dict(ck2["optimizer"])  # torch accepts it as its own
        assert topt.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
        m2 = build(cfg, O.synth_params(cfg, seed=24, std=0.08))
        opt2 = FusedAdam(m2, lr=1.0)
        load_checkpoint(m2, path, opt2, resume=True)
        assert opt2._step == opt._step and torch.equal(opt2._m, opt._m) and torch.equal(opt2._v, opt._v)
        assert m2.step_seed == m.step_seed


def test_grad_reducer_on_the_hip_engine_nccl():
    """GradReducer attached to the HIP model with an RCCL (nccl) process group over the visible devices (world 1 on the
    test box: the collectives run, degenerate): gradients identical to the run without a reducer; the launched spans are
    ordered, contiguous and cover the flat buffer, cut at the engine's real bucket keys; the reducer survives an engine
    rebuild (load_state_dict / set_answer_embeddings after attach -- the reference's call order); accumulate() holds the
    exchange over several backward passes (mc.py)."""
    import socket

    import torch.distributed as dist

    from frozenbilm_amd.parallel import GradReducer

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=81, std=0.05, ln_jitter=0.1)
    batch = to_dev(synth_batch(cfg, B=4, L=50, seed=3))
    m0 = build(cfg, P)
    m0(**batch).loss.backward()
    want = {n: p.grad.clone() for n, p in m0.named_parameters() if p.requires_grad}
    created = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
        created = True
    try:
        m = build(cfg, P)
        red = GradReducer.attach(m, min_bucket_elems=1, overlap="backward")
        eng0 = m.engine()
        m.load_state_dict(P, strict=False)  # invalidates the engine AFTER the reducer was attached
        m.to(DEV)
        eng = m.engine()
        assert eng is not eng0 and eng.reducer is red and red.flat_grad.data_ptr() == eng.flat_grad.data_ptr()
        red.world = 2  # force the collective + 1/world path on the single rank: all-reduce over 1 rank is the identity
        m(**batch).loss.backward()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert torch.allclose(p.grad, want[n] * 0.5, rtol=1e-6, atol=0), n  # SUM over 1 rank, then / world(=2)
        spans = sorted(red.last_launched)  # (stages may become final out of order: the repeated last layer)
        assert spans[0][0] == 0 and spans[-1][1] == eng.flat_grad.numel()
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        ends = set(eng.bucket_ends.values())
        assert all(e in ends for _, e in spans), "buckets must be cut at the engine's backward stages"
        assert len(spans) >= cfg.num_hidden_layers + 2
        # several backward passes feed one exchange
        red.world = 1
        for p in m.parameters():
            p.grad = None
        with red.accumulate():
            m(**batch).loss.backward()
            assert not red.launched and not red.pending
            m(**batch).loss.backward()
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert torch.allclose(p.grad, 2 * want[n], rtol=1e-3, atol=1e-7), n
        assert red.last_launched == [(0, eng.flat_grad.numel())]
    finally:
        if created:
            dist.destroy_process_group()


def _run_two_ranks(tmp_path, overlap, config):
    import os
    import socket
    import subprocess
    import sys

    world = 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out_file = str(tmp_path / f"dp_{overlap}_{config}.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dp_worker.py"), out_file, overlap, config],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p_ in procs:
        try:
            o, _ = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p_.kill()
            o, _ = p_.communicate()
        logs.append(o)
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(logs)
    print("\n".join(l for lg in logs for l in lg.splitlines() if "[dp_worker]" in l))
    return torch.load(out_file)


def _single_process_reference(cfg, P, L, engine_options=None):
    m = build(cfg, P, engine_options=engine_options)
    batch = synth_batch(cfg, B=4, L=L, seed=9)
    m.zero_grad(set_to_none=False)
    losses = []
    for r in range(2):
        out = m(**{k: v[2 * r:2 * r + 2].to(DEV) for k, v in batch.items()})
        out.loss.backward()
        losses.append(out.loss.item())
    return {n: p.grad.float().cpu() / 2 for n, p in m.named_parameters() if p.requires_grad}, losses


@pytest.mark.parametrize("overlap", ["backward", "attention_windows", "after"])
def test_two_rank_data_parallel_equivalence_on_the_hip_engine(tmp_path, overlap):
    """Two processes, one rank each, over RCCL (nccl backend, one GPU per rank) when the box has two GPUs and over gloo
    with both ranks on cuda:0 otherwise: the gradients left in p.grad after backward + GradReducer.finish() equal the
    single-process gradients of the same four samples under the DDP loss convention (mean over ranks of the per-rank mean
    loss) -- same kernels, same inputs, so to fp32 rounding -- on every rank, over two consecutive steps, for each
    placement of the collectives (GradReducer.overlap): from inside the backward pipeline at every final stage, only inside
    the attention-backward windows, or once after backward."""
    got = _run_two_ranks(tmp_path, overlap, "tiny")
    assert got["ranks_agree"] and got["world"] == 2 and got["covers"] and got["overlap"] == overlap
    # the logged loss of a loop step travels with the first bucket: rank-averaged on the host, no collective of its own
    assert got["loss_rides"] and got["extra_collectives_for_the_loss"] == 0, got
    # ("attention_windows" at the tiny size: its six adapters never fill a gradient group, every stage becomes final after
    #  the last window -- one collective; the xlarge-dimension test below sees windows that carry buckets)
    want_n = {"backward": lambda n: n >= 3, "after": lambda n: n == 1, "attention_windows": lambda n: n >= 1}[overlap]
    assert want_n(got["collectives"]), got["launch_order"]
    print(f"backend {got['backend']} (RCCL ranks: {got['rccl_ranks']}), {got['collectives']} bucket collectives per step")
    cfg = _tiny_cfg()
    want, losses = _single_process_reference(cfg, O.synth_params(cfg, seed=41, std=0.05, ln_jitter=0.1), 60)
    assert abs(got["losses"][0] - losses[0]) < 1e-6 and abs(got["losses"][1] - losses[0]) < 1e-6  # rank 0's own loss, both steps
    worst = max(_rel_fro(got["grads"][n], want[n]) for n in want)
    print(f"worst relative difference reduced-vs-single-process: {worst:.2e}")
    assert worst < 1e-5, worst


def test_two_rank_data_parallel_with_packed_rows(tmp_path):
    """model.packed_rows under data parallelism: every rank drops the padding rows of its own shard (the ranks process
    different row counts); the reduced gradients equal the single-process gradients of the padded grid"""
    got = _run_two_ranks(tmp_path, "attention_windows", "tiny+packed")
    assert got["ranks_agree"] and got["world"] == 2 and got["covers"]
    cfg = _tiny_cfg()
    want, losses = _single_process_reference(cfg, O.synth_params(cfg, seed=41, std=0.05, ln_jitter=0.1), 60)
    assert abs(got["losses"][0] - losses[0]) < 1e-5
    worst = max(_rel_fro(got["grads"][n], want[n]) for n in want)
    print(f"worst relative difference reduced (packed rows) vs single-process (padded grid): {worst:.2e}")
    assert worst < 1e-5, worst


def test_two_rank_graph_replay_beside_an_eager_rank(tmp_path):
    """model.training_graphs on both ranks, rank 1 falls back to the eager step (its capture budget is used up): the ranks must
    issue identical collectives -- one all-reduce over the whole flat buffer per step (rank-divergent collective
    patterns) -- and end with the same reduced gradients."""
    got = _run_two_ranks(tmp_path, "attention_windows", "tiny+graphs_mixed")
    assert got["ranks_agree"] and got["world"] == 2 and got["covers"] and got["pattern_same"], got
    assert got["collectives"] == 1 and got["overlap"] == "after", got


@pytest.mark.slow
def test_two_rank_data_parallel_at_xlarge_dimensions(tmp_path):
    """The same check at the true xlarge dimensions (H = 1536, 24 heads, I = 6144, 192-wide adapters; 4 layers, B = 2 per
    rank) with the adapter gradients leaving in groups of 4: the stage buckets become final OUT of order (the repeated last
    layer waits for the second group launch, the layers behind it do not), the collectives start in the attention-backward
    windows -- the default placement -- and the flat buffer is still covered exactly once."""
    got = _run_two_ranks(tmp_path, "attention_windows", "xl4")
    assert got["ranks_agree"] and got["covers"] and got["collectives"] >= 2, got["launch_order"]
    order = got["launch_order"]
    assert order != sorted(order), f"expected an out-of-order bucket launch, got {order}"
    print(f"backend {got['backend']} (RCCL ranks: {got['rccl_ranks']}), launch order {order}")
    cfg = O.OracleConfig()
    cfg.num_hidden_layers, cfg.vocab_size = 4, 4096
    want, _ = _single_process_reference(cfg, O.synth_params(cfg, seed=41, std=0.02, ln_jitter=0.1), 96, engine_options={"dw_group": 4})
    worst = max(_rel_fro(got["grads"][n], want[n]) for n in want)
    print(f"worst relative difference reduced-vs-single-process at xlarge dims: {worst:.2e}")
    assert worst < 1e-5, worst


def test_operands_follow_hidden_parameter_writes_unless_frozen():
    """an inference forward after a write the engine cannot see (through `.data`) must run on the new weights --
    operands are rebuilt on every forward; only inside `model.weights_frozen()` (the evaluate loops) is the rebuild skipped,
    and there FusedAdam.step / in-place updates through the parameter object are still seen."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=41, std=0.05, ln_jitter=0.1)
    m = build(cfg, P, train=False)
    batch = to_dev(synth_batch(cfg, B=2, L=24, seed=3))
    name = "deberta.encoder.layer.1.output.adapter.up.weight"
    p = m.get_param(name)
    with torch.no_grad():
        l0 = m(**batch).loss.item()
        p.data.mul_(3.0)  # hidden write: neither params_version nor the autograd version counter moves
        l1 = m(**batch).loss.item()
        assert abs(l1 - l0) > 1e-4, "the forward still ran on the old adapter weights"
        with m.weights_frozen():
            l2 = m(**batch).loss.item()  # first forward in the scope rebuilds
            assert abs(l2 - l1) < 1e-5  # (the loss is summed with atomics: equal up to the order of its terms)
            p.data.mul_(1.0 / 3.0)  # breaks the promise: the scope keeps the packed operands
            assert abs(m(**batch).loss.item() - l1) < 1e-5
            p.mul_(1.0)  # a write through the parameter object bumps its version counter: seen
            l3 = m(**batch).loss.item()
            assert abs(l3 - l0) < 1e-5
        assert abs(m(**batch).loss.item() - l0) < 1e-5


def test_delayed_loss_check_gives_the_same_epoch_statistics():
    """`args.delayed_loss_check`: the loss of step i is logged when step i+1 calls (asynchronous copy to pinned memory);
    the epoch's averaged statistics and the parameter updates equal the reference-order loop's."""
    import types

    from frozenbilm_amd import main as P_main
    from frozenbilm_amd.optim import FusedAdam
    from tests.downstream_fixtures import ListLoader, StubTokenizer

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=43, std=0.05, ln_jitter=0.1)
    tok = StubTokenizer(cfg.vocab_size)
    g = torch.Generator().manual_seed(9)
    batches = []
    for i in range(4):
        ids = torch.randint(5, cfg.vocab_size, (3, 20), generator=g)
        batches.append(dict(video=torch.randn(3, cfg.max_feats, cfg.features_dim, generator=g), video_len=torch.tensor([cfg.max_feats, 2, 1]),
                            text=[" ".join(str(int(t)) for t in row) for row in ids], qid=[3 * i, 3 * i + 1, 3 * i + 2]))
    stats_out, finals = [], []
    for delayed in (False, True):
        m = build(cfg, P, train=True)
        opt = FusedAdam(m, lr=1e-3)
        args = types.SimpleNamespace(max_tokens=64, mlm_prob=0.15, print_freq=100, epochs=1, lr=1e-3, schedule="",
                                     fraction_warmup_steps=0.1, delayed_loss_check=delayed)
        torch.manual_seed(1234)  # mask_tokens draws from the default CPU generator
        m.step_seed = 0
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            st = P_main.train_one_epoch(m, tok, ListLoader(batches), opt, torch.device(DEV), 0, args, max_norm=1.0)
        stats_out.append(st)
        finals.append(m.get_param("deberta.embeddings.linear_video.weight").detach().clone())
    assert set(stats_out[0]) == set(stats_out[1])
    for k in stats_out[0]:
        assert abs(stats_out[0][k] - stats_out[1][k]) < 1e-5, (k, stats_out)
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("set_to_none", [False, True])
def test_graphed_training_step_equals_the_eager_step(set_to_none):
    """model.training_graphs: forward and backward of the MLM step replayed as two hipGraphs (train_graph.py).  Same kernels,
    same inputs, same seeds (per-site constants + the device word rewritten before every replay): three optimizer steps on
    three different batches leave exactly the parameters the eager loop leaves; dropout is live and differs between steps;
    `.logits` of a graphed step are filled on access; a second forward before backward is refused; an eval forward and a
    shape change in between do not disturb the captured graphs.  set_to_none=True: the standard `opt.zero_grad()` idiom before
    the forward -- every p.grad is None when a new shape is captured, and the eager warm-up step in front of the capture must
    leave them None (it used to leave views of stale gradients behind, which the first replay accumulated onto)."""
    from frozenbilm_amd.optim import FusedAdam

    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=47, std=0.05, ln_jitter=0.1)
    # (the third batch has another shape: its capture -- and the eager warm-up step in front of it -- happens while the flat
    #  gradient buffer still holds the gradients of step 2)
    batches = [to_dev(synth_batch(cfg, B=4, L=40 if i < 2 else 56, seed=60 + i)) for i in range(3)]
    other = to_dev(synth_batch(cfg, B=2, L=24, seed=70))
    results = []
    for graphs in (False, True):
        torch.manual_seed(321)
        m = build(cfg, P, train=True)
        m.training_graphs = graphs
        opt = FusedAdam(m, lr=1e-3, betas=(0.9, 0.95))
        losses = []
        for i, b in enumerate(batches):
            if set_to_none:
                for p_ in m.parameters():
                    p_.grad = None
            else:
                opt.zero_grad(set_to_none=False)
            out = m(**b)
            if graphs and i == 1:
                with pytest.raises(RuntimeError):
                    m(**b)  # the graph owns one set of activations
                lg = out.logits  # filled on access, from the replay's head input
                assert lg.shape == (4, cfg.max_feats + 40, cfg.vocab_size) and torch.isfinite(lg).all()
            out.loss.backward()
            opt.step(clip_max_norm=1.0)
            losses.append(out.loss.item())
            if i == 0:  # an eval forward and a step of another shape in between
                m.eval()
                with torch.no_grad():
                    m(**other)
                m.train()
        results.append((losses, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, m.step_seed))
        if graphs:
            assert len(m.__dict__.get("_train_graphs", {})) == 2
    (l0, p0, s0), (l1, p1, s1) = results
    assert s0 == s1 == 3
    assert all(abs(a - b) < 1e-6 for a, b in zip(l0, l1)), (l0, l1)
    assert len(set(round(x, 6) for x in l0)) == 3
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n


def test_graphed_training_step_soak_300_replays():
    """model.training_graphs over 300 replayed optimizer steps at the true xlarge dimensions (4 layers, B = 8, S = 138: the real
    8-phase GEMM / attention / adapter kernels, split GEMM launches with their aux-stream branch included): the flat gradient
    buffer is finite after EVERY backward replay (counted on the device, read back once per 50 steps so that the replays run
    back to back as in a training loop) and every 100th replay equals the eager step from the same state bit for bit.  Round 5
    saw about one non-finite replay in 500 while the zero fills of accumulation targets were memset nodes of the graphs; with
    kernel-only graphs (fbl_zero) the full-size soak -- tools/soak_graphs.py, 24 layers, B = 32: 6 500 replays, 26 eager
    comparisons, profiles/r06_soak_graphs*.json -- and this short one are clean.  The step it replays: reference main.py:59-90."""
    from frozenbilm_amd.optim import FusedAdam

    cfg = O.OracleConfig()
    cfg.num_hidden_layers = 4
    cfg.vocab_size = 4096
    P = O.synth_params(cfg, seed=71)
    torch.manual_seed(99)
    m = build(cfg, P, train=True)
    opt = FusedAdam(m, lr=3e-5, betas=(0.9, 0.95))
    batch = to_dev(synth_batch(cfg, B=8, L=128, seed=5))
    m.training_graphs = True
    n = 300
    nf = torch.zeros(n, dtype=torch.int32, device=DEV)
    mism = []
    for i in range(n):
        seed_before = m.step_seed
        opt.zero_grad(set_to_none=False)
        out = m(**batch)
        out.loss.backward()
        g = m.engine().flat_grad
        nf[i] = (~torch.isfinite(g)).sum() + (~torch.isfinite(out.loss.detach())).to(torch.int32)
        if i % 100 == 99:
            g_rep, l_rep = g.clone(), out.loss.detach().clone()
            m.training_graphs = False
            m.step_seed = seed_before
            opt.zero_grad(set_to_none=False)
            o2 = m(**batch)
            o2.loss.backward()
            if not (torch.equal(g_rep, m.engine().flat_grad) and torch.equal(l_rep, o2.loss.detach())):
                mism.append(i)
            m.training_graphs = True
            g.copy_(g_rep)
        opt.step(clip_max_norm=0.1)
        if i % 50 == 49:
            bad = nf[:i + 1].cpu()
            assert int(bad.sum()) == 0, f"non-finite gradient in replay {int(torch.nonzero(bad)[0])}"
    assert not mism, mism
    assert m.__dict__.get("_train_graph_captures", 0) == 1


def test_inference_shortcuts_change_nothing():
    """Inference forwards inside model.weights_frozen() keep the per-layer position projections across forwards, reuse the
    key/value projection of the first enhanced-mask-decoder pass in the second and, when only a loss is asked for, run the
    prediction head on the labelled rows: same loss, same logits (filled on access) as the general path; a parameter update
    in between is seen; the [MASK]-row path (logit_rows) agrees as well."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=53, std=0.05, ln_jitter=0.1)
    m = build(cfg, P, train=False)
    batch = to_dev(synth_batch(cfg, B=3, L=30, seed=12))
    S = cfg.max_feats + 30
    rows = torch.tensor([0 * S + 12, 1 * S + 15, 2 * S + 20], device=DEV)
    feed = {k: v for k, v in batch.items() if k != "labels"}
    with torch.no_grad():
        ref = m(**batch)
        ref_loss, ref_logits = ref.loss.item(), ref.logits.clone()
        ref_rows = m(**feed, logit_rows=rows).logits.clone()
        with m.weights_frozen():
            for it in range(3):  # first forward fills the position cache, the next ones use it
                out = m(**batch)
                assert abs(out.loss.item() - ref_loss) < 1e-5
            assert len(m.engine()._pos_cache) == cfg.num_hidden_layers
            assert (out.logits - ref_logits).abs().max().item() < 1e-5  # head on every row, on access
            assert (m(**feed, logit_rows=rows).logits - ref_rows).abs().max().item() < 1e-5
            # an update through the parameter object is seen: cache dropped, new values used
            p = m.get_param("deberta.encoder.LayerNorm.weight")
            p.mul_(1.5)
            l2 = m(**batch).loss.item()
            p.mul_(1.0 / 1.5)
        general = m(**batch).loss.item()  # outside the scope: the general path
        p.mul_(1.5)
        l2_general = m(**batch).loss.item()
        p.mul_(1.0 / 1.5)
    assert abs(general - ref_loss) < 1e-5 and abs(l2 - l2_general) < 1e-5 and abs(l2 - ref_loss) > 1e-6


def _packed_vs_grid(cfg, P, batch, tol_logits, tol_grad):
    """the same model with and without `packed_rows` on one ragged batch (eval mode: no dropout streams to differ)"""
    m = build(cfg, P, train=False)
    dev_batch = to_dev(batch)
    B, Lt = batch["input_ids"].shape
    S = cfg.max_feats + Lt
    out_g = m(**dev_batch, output_hidden_states=True)
    out_g.loss.backward()
    grads_g = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}
    logits_g = out_g.logits.detach().clone()
    hid_g = [h.clone() for h in out_g.hidden_states]
    m.zero_grad(set_to_none=True)
    m.packed_rows = True
    out_p = m(**dev_batch, output_hidden_states=True)
    run = out_p._run
    assert run.pk is not None and run.N == run.pk.n < B * S, "the batch was not packed"
    out_p.loss.backward()
    # rows that exist: every position up to the last valid / labelled one of its sample
    exist = torch.zeros(B * S, dtype=torch.bool, device=DEV)
    exist[run.pk.sel] = True
    full_mask = torch.cat([dev_batch["video_mask"], dev_batch["attention_mask"]], 1).bool().view(-1)
    assert bool((exist | ~full_mask).all()), "a valid position lost its row"
    assert abs(out_p.loss.item() - out_g.loss.item()) < 1e-4 * max(1.0, abs(out_g.loss.item()))
    lp = out_p.logits.detach().reshape(B * S, -1)
    lg = logits_g.reshape(B * S, -1)
    d_log = (lp[exist] - lg[exist]).abs().max().item()
    assert d_log < tol_logits, d_log
    assert float(lp[~exist].abs().max().item() if bool((~exist).any()) else 0.0) == 0.0  # dropped positions read as zero
    for hp, hg in zip(out_p.hidden_states, hid_g):
        hp, hg = hp.reshape(B * S, -1), hg.reshape(B * S, -1)
        assert (hp[exist] - hg[exist]).abs().max().item() < tol_logits
        assert float(hp[~exist].abs().max().item()) == 0.0
    worst = 0.0
    for n, p in m.named_parameters():
        if p.requires_grad:
            ref = grads_g[n].float()
            worst = max(worst, (p.grad.float() - ref).norm().item() / max(ref.norm().item(), 1e-12))
    assert worst < tol_grad, worst
    print(f"packed vs grid: rows {run.N}/{B * S}, logits max-abs diff {d_log:.2e}, worst gradient Frobenius diff {worst:.2e}")
    return m, run


def test_packed_rows_equal_the_padded_grid_tiny():
    """model.packed_rows: a ragged batch without its trailing padding rows gives the loss, the logits / hidden states at
    every position that has a row and the gradients of the padded grid (tiny config: 3 layers incl. the convolution, the
    enhanced mask decoder and the position-table gradients); positions without a row read as zero."""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=61, std=0.05, ln_jitter=0.1)
    for B, Lt, seed in ((5, 90, 3), (2, 33, 4), (7, 130, 5)):
        _packed_vs_grid(cfg, P, synth_batch(cfg, B=B, L=Lt, seed=seed), tol_logits=2e-2, tol_grad=3e-2)


def test_packed_rows_against_the_oracle_and_on_selected_rows():
    """the packed forward / backward against the fp32 CPU oracle (same bounds as the padded path), and the inference
    forward on selected rows (logit_rows, the downstream loops) packed vs padded"""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=62, std=0.05, ln_jitter=0.1)
    batch = synth_batch(cfg, B=6, L=101, seed=8)
    for k, v in P.items():
        v.requires_grad_(O.is_trainable(k))
    ref = O.forward(P, cfg, **batch)
    ref["loss"].backward()
    m = build(cfg, P)
    m.packed_rows = True
    out = m(**to_dev(batch))
    out.loss.backward()
    assert out._run.pk is not None
    assert abs(out.loss.item() - ref["loss"].item()) < 2e-2
    sel = out._run.pk.sel.cpu()
    lg = out.logits.detach().float().cpu().reshape(-1, ref["logits"].shape[-1])
    assert (lg[sel] - ref["logits"].detach().reshape(lg.shape)[sel]).abs().max().item() < 5e-2
    for name, p in m.named_parameters():
        if p.requires_grad:
            r = P[name].grad
            fro = (p.grad.float().cpu() - r).norm().item() / max(r.norm().item(), 1e-9)
            assert fro < (0.25 if "adapter.down" in name else 6e-2), (name, fro)
    # [MASK]-row inference: rows given as indices into the padded grid, whatever the layout underneath
    S = cfg.max_feats + 101
    feed = {k: v for k, v in to_dev(batch).items() if k != "labels"}
    rows = torch.tensor([b * S + cfg.max_feats + 1 for b in range(6)], device=DEV)
    with torch.no_grad():
        packed = m(**feed, logit_rows=rows).logits.clone()
        m.packed_rows = False
        grid = m(**feed, logit_rows=rows).logits
    assert packed.shape == grid.shape and (packed - grid).abs().max().item() < 2e-2
    # a loss-only inference forward (main.evaluate) whose logits are read after all: head on the labelled rows first, on
    # every packed row at the access
    dev_batch = to_dev(batch)
    with torch.no_grad(), m.weights_frozen():
        ref_out = m(**dev_batch)
        ref_loss, ref_logits = ref_out.loss.item(), ref_out.logits.clone()
        m.packed_rows = True
        out2 = m(**dev_batch)
        assert out2._run.pk is not None and abs(out2.loss.item() - ref_loss) < 1e-5
        sel_d = out2._run.pk.sel
        got = out2.logits.reshape(-1, ref_logits.shape[-1])
        assert (got[sel_d] - ref_logits.reshape(got.shape)[sel_d]).abs().max().item() < 1e-5


def test_packed_rows_training_step_is_reproducible_and_finite():
    """training mode (dropout live, keyed by packed row): two models stepped from the same state draw the same masks and
    give the same gradients bit for bit; losses and gradients are finite; the loss stays close to the padded run's (different mask stream)"""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=63, std=0.05, ln_jitter=0.1)
    batch = to_dev(synth_batch(cfg, B=6, L=77, seed=9))
    losses, grads = [], []
    for packed in (True, True, False):
        m = build(cfg, P, train=True)
        m.packed_rows = packed
        out = m(**batch)
        out.loss.backward()
        assert (out._run.pk is not None) == packed
        losses.append(out.loss.item())
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.requires_grad]).clone())
        assert torch.isfinite(grads[-1]).all() and torch.isfinite(out.loss)
    assert losses[0] == losses[1] and torch.equal(grads[0], grads[1])  # (the loss is a fixed-order fold: no atomics)
    assert abs(losses[0] - losses[2]) < 0.2  # same model, another dropout stream


@pytest.mark.slow
def test_packed_rows_equal_the_padded_grid_at_xlarge_dimensions():
    """the same at the true xlarge dimensions (4 layers): the 8-phase GEMMs, 24 heads, 192-wide adapters"""
    cfg = O.OracleConfig()
    cfg.num_hidden_layers, cfg.vocab_size = 4, 4096
    P = O.synth_params(cfg, seed=64, std=0.02, ln_jitter=0.1)
    _packed_vs_grid(cfg, P, synth_batch(cfg, B=8, L=256, seed=11), tol_logits=3e-2, tol_grad=3e-2)


def test_packed_rows_gradient_through_the_logits():
    """a loss the caller builds on the returned logits (downstream fine-tuning style) next to the internal MLM loss: the
    gradient handed in on the [B, S, V] grid reaches the packed rows"""
    cfg = _tiny_cfg()
    P = O.synth_params(cfg, seed=65, std=0.05, ln_jitter=0.1)
    batch = to_dev(synth_batch(cfg, B=4, L=70, seed=13))
    valid = torch.cat([batch["video_mask"], batch["attention_mask"]], 1).bool()
    w = torch.randn(4, cfg.max_feats + 70, cfg.vocab_size, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    w = w * valid[..., None] * 1e-3
    grads = []
    for packed in (False, True):
        m = build(cfg, P)
        m.packed_rows = packed
        out = m(**batch)
        (out.loss + (out.logits * w).sum()).backward()
        assert (out._run.pk is not None) == packed
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad})
    for n in grads[0]:
        ref = grads[0][n].float()
        assert (grads[1][n].float() - ref).norm().item() <= 1e-4 * max(ref.norm().item(), 1e-9), n
