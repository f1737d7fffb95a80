"""Per-stage reference goldens on the GPU: the inputs the reference's own modules were fed when G1 (Adapter), G3
(DisentangledSelfAttention) and G4 (DebertaV2Layer, ConvLayer) were captured go STRAIGHT to the HIP kernels / engine
stages, and the outputs are compared with what the reference returned (model/adapter.py:33-45,
model/deberta.py:717-818, :351-419) -- no oracle, no third restatement in between.  Tolerances: bf16 MFMA operands with
fp32 accumulation and an fp32 residual stream (5e-2 max-abs as for the logits; measured values are printed).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import deberta_oracle as O  # noqa: E402  (only synth_params: the goldens' weights are a seed recipe)
from tests.golden.make_goldens import _tiny_cfg  # noqa: E402
from tests.test_gpu_model import build  # noqa: E402

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32


def _engine_run(m, mask):
    """an Engine + a Run carrying what the layer stages read (mask, klen, dispatch order, rel-embedding table)"""
    from frozenbilm_amd.engine import Run

    eng = m.engine()
    eng.refresh_trainable_operands()
    B, S = mask.shape
    run = Run(B=B, S=S, T=0, Lt=S, train=False, save=False, seed_base=0, p_hid=0.0, p_att=0.0, p_ad=0.0)
    run.mask = mask.to(DEV).to(torch.int32).contiguous().view(-1)
    run.mask_i32 = run.rowmask = run.mask
    run.mask_f = run.mask.to(F32)
    pos1 = torch.arange(1, S + 1, device=DEV, dtype=torch.int32)
    run.klen = (run.mask.view(B, S) * pos1).amax(1).to(torch.int32).contiguous()
    run.border = torch.argsort(run.klen, descending=True, stable=True).to(torch.int32).contiguous()
    r, _ = eng._ln(run, "deberta.encoder.LayerNorm", y=eng.rel_emb, resid=None, N=eng.span2, want_f32=True)
    run.rel_norm, run.R32 = r.norm, r.plain
    return eng, run, r


def _stream(eng, x):
    """[B,S,H] fp32 activations as the engine's stream type (bf16 operand with the table tail + fp32 value)"""
    from frozenbilm_amd.engine import Stream

    H = x.shape[-1]
    x32 = x.reshape(-1, H).to(DEV).float().contiguous()
    full = torch.zeros(x32.shape[0] + eng.span2, H, dtype=BF16, device=DEV)
    full[: x32.shape[0]] = x32
    return Stream(bf16=full[: x32.shape[0]], plain=x32, full=full)


def test_g1_adapter_module(golden):
    from frozenbilm_amd.model.adapter import Adapter

    g = golden("G1_adapter")
    ad = Adapter(8, 128, dropout=0.1).to(DEV).eval()
    with torch.no_grad():
        for k in ("down.weight", "down.bias", "up.weight", "up.bias"):
            ad.get_parameter(k).copy_(g["w." + k].to(DEV))
        y = ad(g["x"].to(DEV))
    err = (y.float().cpu() - g["y"]).abs().max().item()
    print(f"G1 adapter max-abs err {err:.2e}")
    assert y.shape == g["y"].shape and err < 2e-2, err


@pytest.mark.parametrize("S", [37, 266])
def test_g3_attention_stage(golden, S):
    """QKV + position projections (one GEMM) and the fused attention kernel on the reference module's inputs"""
    from frozenbilm_amd import lib as L

    g = golden("G3_attention")
    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1))
    hidden, qs, mask = g[f"hidden_{S}"], g[f"qs_{S}"], g[f"mask_{S}"]
    eng, run, r = _engine_run(m, mask)
    assert (r.plain.cpu() - g["rel_emb"]).abs().max().item() < 1e-4  # LayerNorm of the rel-embedding table (fp32 kernel)
    B, H, nh, N, P_ = run.B, eng.H, eng.nh, run.B * S, eng.span2
    Sp = (S + 63) // 64 * 64
    W = eng.Lw[1]
    for name, q_in in (("ctx", None), ("ctxq", qs)):
        kv = _stream(eng, hidden)
        kv.full[N:].copy_(r.bf16)
        qkv = torch.empty(N + P_, 3 * H, dtype=BF16, device=DEV)
        if q_in is None:
            L.gemm(kv.full, W["Wqkv"], bias=W["bqkv"], out_bf16=qkv)
        else:
            q = _stream(eng, q_in)
            q.full[N:].copy_(r.bf16)
            L.gemm(q.full, W["Wqkv"][:H], bias=W["bqkv"][:H], out_bf16=qkv[:, :H])
            L.gemm(kv.full, W["Wqkv"][H:], bias=W["bqkv"][H:], out_bf16=qkv[:, H:])
        ctx = torch.empty(N, H, dtype=BF16, device=DEV)
        lse = torch.empty(B, nh, S, dtype=F32, device=DEV)
        L.disent_attn_fwd(qkv[:N, :H], qkv[:N, H:2 * H], qkv[:N, 2 * H:], qkv[N:, H:2 * H], qkv[N:, :H], eng.relidx(S),
                          run.mask_i32, 1.0 / math.sqrt(64 * 3), ctx, lse, B, S, Sp, nh, P_, klen=run.klen, border=run.border,
                          lin=eng.lin_span)
        ref = g[f"{name}_{S}"]
        got = ctx.float().cpu().view(B, S, H)
        err = (got - ref).abs().max().item()
        print(f"G3 {name} S={S}: max-abs err {err:.2e} (ref max {ref.abs().max().item():.2f})")
        assert err < 2e-2, (name, err)
        assert (got[0, S - 5:] == 0).all()  # padded query rows: XSoftmax gives exact zeros


@pytest.mark.parametrize("S", [37, 266])
def test_g3b_attention_backward_stage(golden, S):
    """The attention backward kernels (prep, dS / dV, the two shear passes, the position-table products) on the reference
    module's inputs and a seeded upstream gradient, against the gradients the REFERENCE's own backward left on the outputs of
    query_proj / key_proj / value_proj -- token rows: dq, dk, dv; relative-position rows: dPQ, dPK (G3b; hooks on the
    reference module, tests/golden/make_goldens.py).  Until round 4 this stage was only checked against the builder's own
    restatement (tests/gpu_refs.py) and, in aggregate, through the model-level gradient goldens."""
    import types

    from frozenbilm_amd import lib as L
    from frozenbilm_amd.attn_bwd import disent_attn_bwd

    g, gb = golden("G3_attention"), golden("G3b_attention_backward")
    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=3, std=0.05, ln_jitter=0.1))
    hidden, qs, mask, dy = g[f"hidden_{S}"], g[f"qs_{S}"], g[f"mask_{S}"], gb[f"dy_{S}"]
    eng, run, r = _engine_run(m, mask)
    B, H, nh, N, P_ = run.B, eng.H, eng.nh, run.B * S, eng.span2
    Sp = (S + 63) // 64 * 64
    W = eng.Lw[1]
    sl = (lambda t: t) if S == 37 else (lambda t: t[:, ::3])
    for tag, q_in in (("", None), ("q", qs)):
        kv = _stream(eng, hidden)
        kv.full[N:].copy_(r.bf16)
        qkv = torch.empty(N + P_, 3 * H, dtype=BF16, device=DEV)
        if q_in is None:
            L.gemm(kv.full, W["Wqkv"], bias=W["bqkv"], out_bf16=qkv)
        else:
            q = _stream(eng, q_in)
            q.full[N:].copy_(r.bf16)
            L.gemm(q.full, W["Wqkv"][:H], bias=W["bqkv"][:H], out_bf16=qkv[:, :H])
            L.gemm(kv.full, W["Wqkv"][H:], bias=W["bqkv"][H:], out_bf16=qkv[:, H:])
        ctx = torch.empty(N, H, dtype=BF16, device=DEV)
        lse = torch.empty(B, nh, S, dtype=F32, device=DEV)
        L.disent_attn_fwd(qkv[:N, :H], qkv[:N, H:2 * H], qkv[:N, 2 * H:], qkv[N:, H:2 * H], qkv[N:, :H], eng.relidx(S),
                          run.mask_i32, 1.0 / math.sqrt(64 * 3), ctx, lse, B, S, Sp, nh, P_, klen=run.klen, border=run.border,
                          lin=eng.lin_span)
        sv = types.SimpleNamespace(qkv=qkv[:N], pqk=qkv[N:, : 2 * H], ctx=ctx, lse=lse, seed_att=0)
        dctx = dy.reshape(N, H).to(DEV).to(BF16).contiguous()
        dqkv = torch.full((N, 3 * H), float("nan"), dtype=BF16, device=DEV)
        dpqk = torch.empty(P_, 2 * H, dtype=BF16, device=DEV)
        disent_attn_bwd(eng, run, sv, dctx, dqkv, dpqk)
        torch.cuda.synchronize()
        got = dqkv.float().cpu().view(B, S, 3 * H)
        worst = {}
        for name, lo in (("dq", 0), ("dk", H), ("dv", 2 * H)):
            ref = gb[f"{name}{tag}_{S}"].float()
            out = sl(got[:, :, lo:lo + H])
            worst[name] = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        for name, lo in (("dpq", 0), ("dpk", H)):
            ref = gb[f"{name}{tag}_{S}"].float()
            out = dpqk[:, lo:lo + H].float().cpu()
            worst[name] = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        print(f"G3b S={S} query_states={'yes' if tag else 'no'}: max-abs error / max |reference| per gradient: "
              + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
        # bf16 operands (q, k, v, dO, P, dS all rounded to bf16 on the way) against the fp32 reference
        assert all(v < 3e-2 for v in worst.values()), worst
        assert torch.isfinite(got).all()  # every row of dq / dk / dv written (padded rows included)


def test_g4_layer_and_conv_stages(golden):
    """the engine's layer stage (attention + adapters + FFN + LayerNorms; encoder form and EMD form with query_states)
    and conv stage on the reference DebertaV2Layer / ConvLayer inputs"""
    g = golden("G4_layer_conv")
    cfg = _tiny_cfg()
    m = build(cfg, O.synth_params(cfg, seed=4, std=0.05, ln_jitter=0.1))
    hidden, qs, mask = g["hidden"], g["qs"], g["mask"]
    eng, run, r = _engine_run(m, mask)
    B, S, H = hidden.shape
    with torch.no_grad():
        y = eng._layer_fwd(run, 2, _stream(eng, hidden), None, r.bf16)
        yq = eng._layer_fwd(run, 2, _stream(eng, hidden), _stream(eng, qs), r.bf16)
        yc = eng._conv_fwd(run, _stream(eng, hidden), _stream(eng, qs))
    for name, s, ref in (("y_layer", y, g["y_layer"]), ("y_layer_q", yq, g["y_layer_q"]), ("y_conv", yc, g["y_conv"])):
        got = eng._materialize(s).float().cpu().view(B, S, H)
        err = (got - ref).abs().max().item()
        print(f"G4 {name}: max-abs err {err:.2e} (ref max {ref.abs().max().item():.2f})")
        assert err < 5e-2, (name, err)


# ------------------------------------------------------------------------------------------------ ABI contract on the GPU
def _split_gemm_operands(seed=0):
    g = torch.Generator().manual_seed(seed)
    M, N, K = 8512, 6144, 1536  # FFN-up: three whole rounds of 256x256 tiles + 320 remainder rows (the split launch)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF16).to(DEV)
    Bm = (torch.randn(N, K, generator=g) * 0.05).to(BF16).to(DEV)
    return A, Bm


def test_split_gemm_is_capturable_and_owns_no_stream():
    """fbl_gemm_bf16_nt with a caller-provided aux stream inside a stream capture on a fresh stream: the remainder-row
    launch forks to the aux stream and joins back by events, so the whole call lands in ONE graph; replays reproduce the
    eager result bit for bit (include/fbl.h: "only enqueues work on `stream`")."""
    from frozenbilm_amd import lib as L

    A, Bm = _split_gemm_operands()
    eager = torch.empty(A.shape[0], Bm.shape[0], dtype=BF16, device=DEV)
    saved = dict(L._AUX)
    L.set_aux_stream(None)
    L.gemm(A, Bm, out_bf16=eager)  # no aux stream: remainder rows precede the big tiles on the same stream
    torch.cuda.synchronize()
    ref = A[:512].float() @ Bm.float().t()
    assert (eager[:512].float() - ref).abs().max().item() < 0.05 * ref.abs().max().item()
    aux = torch.cuda.Stream()
    L.set_aux_stream(aux)
    try:
        out = torch.zeros_like(eager)
        L.gemm(A, Bm, out_bf16=out)  # forked remainder, eager
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        cap = torch.cuda.Stream()
        out.zero_()
        graph = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            graph.capture_begin()
            L.gemm(A, Bm, out_bf16=out)
            graph.capture_end()
        torch.cuda.synchronize()
        for _ in range(3):
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager)
    finally:
        L._AUX.clear()
        L._AUX.update(saved)


def test_two_threads_two_streams_share_the_aux_stream():
    """the only process state of the library is the mutex-guarded event pool keyed by (stream, aux_stream): two Python
    threads launching split GEMMs on their own streams with one shared aux stream get correct results every time"""
    import threading

    from frozenbilm_amd import lib as L

    A, Bm = _split_gemm_operands(1)
    ref = torch.empty(A.shape[0], Bm.shape[0], dtype=BF16, device=DEV)
    saved = dict(L._AUX)
    L.set_aux_stream(None)
    L.gemm(A, Bm, out_bf16=ref)
    torch.cuda.synchronize()
    aux = torch.cuda.Stream()
    L.set_aux_stream(aux)
    errs = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            out = torch.empty_like(ref)
            with torch.cuda.stream(st):
                for it in range(6):
                    out.zero_()
                    L.gemm(A, Bm, out_bf16=out)
                    st.synchronize()
                    if not torch.equal(out, ref):
                        errs.append((i, it))
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    try:
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        L._AUX.clear()
        L._AUX.update(saved)
    assert not errs, errs
