"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain torch fp32 reference of the same op.

Operands are pre-rounded to bf16 where the kernel consumes bf16, so the tolerances only cover accumulation order and
bf16 rounding of OUTPUTS (<= 2^-8 relative).  All calls go through the C ABI (frozenbilm_amd.lib -> libfbl.so).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_refs import bf, heads, ref_attention, stats, unheads  # noqa: E402

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def L():
    from frozenbilm_amd import lib

    lib.load()
    assert torch.cuda.is_available()
    return lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def close(got, ref, rtol, atol, name=""):
    ok = torch.allclose(got.float(), ref.float(), rtol=rtol, atol=atol)
    assert ok, stats(name, got.float(), ref.float())


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (200, 130, 128), (1000, 1536, 1536), (333, 192, 1536), (256, 4608, 192),
                                   (130, 6, 64), (4100, 2052, 128), (8512, 1536, 192), (4100, 3584, 128), (8512, 192, 1536),
                                   (2100, 70, 64), (4100, 3500, 256), (4100, 3584, 1536), (8512, 6144, 384), (8512, 1536, 384), (4100, 2052, 256),
                                   (4100, 4608, 256), (4100, 4500, 192), (2500, 8192, 128), (8512, 6144, 1536), (691, 16500, 256)])
# the last four: more than one round of 256x256 tiles (skewed single launch / whole rounds + remainder rows), K = 4, 3, 2 and
# 24 K-tiles (3 and 2: below the 8-phase kernel's shortest pipeline -> 2-stage 256x256 tiles), ragged last tile column / row;
# (4100,35xx,K>=256): the 8-phase 256x256 kernel (gemm8.hip; K = 256 is its shortest pipeline, N = 3500 a ragged last column
# tile); (8512,6144): 8-phase tiles for the whole rounds + 64x128 tiles for the remaining rows (helper stream); (8512,1536,384) and
# (4100,2052,256): 224-row 8-phase tiles (second wave row with three 16-row MFMA tiles per half);
# (4100,2052) and (8512,1536): 224x256 tiles (fewer rounds x area); (4100,3584): 256x256 tiles (238 tiles, one round);
# (8512,192) and (2100,70): 64x128 tiles (tall and narrow: the 128x128 grid would not cover the chip);
# (691,16500): a few hundred rows against a very wide N -- the vocabulary GEMM of the loss on the labelled rows -- on 8-phase tiles
def test_gemm_plain_bias(L, M, N, K):
    A, B = bf(rnd(M, K, seed=1)).to(BF16), bf(rnd(N, K, seed=2, scale=0.05)).to(BF16)
    bias = rnd(N, seed=3)
    ldc = (N + 7) // 8 * 8
    o32 = torch.full((M, ldc), 7.0, dtype=F32, device=DEV)
    o16 = torch.zeros(M, ldc, dtype=BF16, device=DEV)
    L.gemm(A, B, bias=bias, out_f32=o32, out_bf16=o16, N=N)
    ref = A.float() @ B.float().t() + bias
    close(o32[:, :N], ref, 1e-4, 1e-3, "gemm f32")
    close(o16[:, :N], ref, 1e-2, 1e-2, "gemm bf16")
    if ldc > N:
        assert (o32[:, N:] == 7.0).all(), "wrote outside N"


@pytest.mark.parametrize("M,N,K", [(8512, 192, 1536), (4100, 192, 1536), (2050, 96, 1536), (5000, 176, 1536), (9024, 16, 1536), (3000, 192, 192),
                                   (2049, 48, 1536)])
def test_gemm_tall_and_narrow_strided_views(L, M, N, K):
    """Tall-and-narrow launches (the adapter backward's dz = (dy . Wu) (*) gate, autograd of model/adapter.py:38-42): plain and
    gated (AUX_MUL_POS_BF16) epilogues, strided operand / aux / output views as the engine passes them (dy and dz are column
    blocks of one [N, 1792] buffer), row counts that are not a multiple of the tile height, column counts that leave the last tile
    column partly empty; nothing outside the output's columns is written."""
    A_full = bf(rnd(M, K + 256, seed=1)).to(BF16)
    A = A_full[:, :K]  # lda = K + 256
    B = bf(rnd(N, K, seed=2, scale=0.05)).to(BF16)
    base = A.float() @ B.float().t()
    out_full = torch.full((M, N + 40), 3.0, dtype=BF16, device=DEV)
    out = out_full[:, 8:8 + N]
    L.gemm(A, B, out_bf16=out, N=N)
    close(out, base, 1e-2, 1e-2, "narrow plain")
    assert (out_full[:, :8] == 3.0).all() and (out_full[:, 8 + N:] == 3.0).all(), "wrote outside the output columns"
    z_full = bf(rnd(M, N + 16, seed=4)).to(BF16)
    z = z_full[:, :N]
    out_full.fill_(3.0)
    L.gemm(A, B, alpha=1.0 / 0.9, aux=z, aux_kind=L.AUX_MUL_POS_BF16, out_bf16=out, N=N)
    close(out, base / 0.9 * (z.float() > 0), 1e-2, 1e-2, "narrow gated")
    assert (out_full[:, :8] == 3.0).all() and (out_full[:, 8 + N:] == 3.0).all()
    # an asymmetric operand pins the orientation: A = [I | 0] picks the first columns of B
    if K <= M:
        A2 = torch.zeros(M, K, dtype=BF16, device=DEV)
        A2[:K] = torch.eye(K, dtype=BF16, device=DEV)
        Bi = (torch.arange(N * K, device=DEV).view(N, K) % 251).to(BF16)
        o2 = torch.empty(M, N, dtype=BF16, device=DEV)
        L.gemm(A2, Bi, out_bf16=o2)
        assert torch.equal(o2[:K].float(), Bi.float().t()) and (o2[K:] == 0).all()


def test_dropout_sum_over_slices(L):
    """fbl_dropout_sum_f32: out = sum_s dropout_{seed_s}(x[s]) in slice order -- bit-identical to fbl_dropout_f32 per slice followed
    by sequential additions (the per-layer-execution chain it replaces), p = 0 is a plain ordered sum, the device seed word is added
    to every slice's seed."""
    E, n = 7, 512 * 96 + 40
    x = rnd(E, n, seed=11)
    seeds = [(0x1234567 + 0x85EBCA77 * (e + 1)) & 0xFFFFFFFFFFFFFFFF for e in range(E)]
    out = torch.empty(n, dtype=F32, device=DEV)
    L.dropout_sum_f32(x, seeds, 0.1, out)
    ref = torch.zeros(n, dtype=F32, device=DEV)
    tmp = torch.empty(n, dtype=F32, device=DEV)
    for e in range(E):
        L.dropout_f32(x[e], 0.1, seeds[e], out_f32=tmp)
        ref += tmp
    assert torch.equal(out, ref)
    assert 0.05 < (out == 0).float().mean().item() * 0 + ((tmp == 0).float().mean().item()) < 0.15
    L.dropout_sum_f32(x, [0] * E, 0.0, out)
    ref.zero_()
    for e in range(E):
        ref += x[e]
    assert torch.equal(out, ref)
    word = torch.tensor([987654321], dtype=torch.int64, device=DEV)
    with L.seed_word(word):
        L.dropout_sum_f32(x, seeds, 0.1, out)
    out2 = torch.empty_like(out)
    L.dropout_sum_f32(x, [(s_ + 987654321) & 0xFFFFFFFFFFFFFFFF for s_ in seeds], 0.1, out2)
    assert torch.equal(out, out2)
    # a slice that starts at element key0 of the keyed tensor
    k0 = 96 * 37
    full = torch.empty(n, dtype=F32, device=DEV)
    L.dropout_sum_f32(x, seeds, 0.1, full)
    part = torch.empty(n - k0, dtype=F32, device=DEV)
    L.dropout_sum_f32(x[:, k0:].contiguous(), seeds, 0.1, part, key0=k0)
    assert torch.equal(part, full[k0:])


def test_heads_to_rows_and_zero(L):
    """fbl_heads_to_rows_bf16 ([E, nh, rows, 64] fp32 -> column block of [E, rows, 2*nh*64] bf16) and fbl_zero"""
    E, nh, rows = 3, 5, 37
    src = rnd(E, nh, rows, 64, seed=4)
    dst = torch.full((E, rows, 2 * nh * 64), 9.0, dtype=BF16, device=DEV)
    L.heads_to_rows_bf16(src, dst[:, :, nh * 64:])
    ref = src.permute(0, 2, 1, 3).reshape(E, rows, nh * 64).to(BF16)
    assert torch.equal(dst[:, :, nh * 64:], ref) and (dst[:, :, : nh * 64] == 9.0).all()
    t = torch.full((1000, 33), 5.0, device=DEV)
    assert L.zero_(t) is t and (t == 0).all()
    z = L.zeros(7, 3, dtype=BF16, device=DEV)
    assert z.shape == (7, 3) and (z == 0).all()
    # any byte range: unaligned start, ragged tail, neighbours untouched; a large fill (grid-stride path)
    raw = torch.full((4099,), 7, dtype=torch.uint8, device=DEV)
    for off, n in ((0, 4099), (1, 15), (3, 64), (13, 4000), (16, 1), (4098, 1)):
        raw.fill_(7)
        L.zero_(raw[off: off + n])
        assert (raw[off: off + n] == 0).all() and (raw[:off] == 7).all() and (raw[off + n:] == 7).all(), (off, n)
    big = torch.full((50_000_003,), 1.0, device=DEV)
    L.zero_(big[1:])
    assert big[0].item() == 1.0 and not big[1:].any()


def test_gemm_asymmetric_identity(L):
    """A = I catches transposed C writes (guide rule: always test with an asymmetric B)."""
    K = 128
    A = torch.eye(K, dtype=BF16, device=DEV)
    B = (torch.arange(96 * K, device=DEV).view(96, K) % 251).to(BF16)
    o = torch.empty(K, 96, dtype=F32, device=DEV)
    L.gemm(A, B, out_f32=o)
    assert torch.equal(o, B.float().t())


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (4100, 2052, 64), (4100, 3584, 64), (4100, 192, 128), (4100, 3584, 256), (4100, 2052, 256),
                                   (4100, 4608, 320)])
# 128x128, 224x256, 256x256 and 64x128 tiles, 8-phase 256x256, 8-phase 224x256 (plain / residual-add) + its fallbacks,
# a multi-round problem with an odd number of K-tiles (306 tiles of 256x256, K = 320: 2-stage 256x256 tiles)
def test_gemm_epilogues(L, M, N, K):
    A, B = bf(rnd(M, K, seed=1)).to(BF16), bf(rnd(N, K, seed=2, scale=0.1)).to(BF16)
    bias = rnd(N, seed=3)
    rows = (torch.arange(M, device=DEV) % 3 != 0).float()
    base = A.float() @ B.float().t()
    # gelu + pre-activation copy + rowscale
    o16 = torch.empty(M, N, dtype=BF16, device=DEV)
    pre = torch.empty(M, N, dtype=BF16, device=DEV)
    L.gemm(A, B, bias=bias, rowscale=rows, act=L.ACT_GELU, out_bf16=o16, out_pre=pre)
    p = (base + bias) * rows[:, None]
    close(pre, p, 1e-2, 1e-2, "pre")
    close(o16, F.gelu(p), 1e-2, 1e-2, "gelu")
    # relu
    L.gemm(A, B, bias=bias, act=L.ACT_RELU, out_bf16=o16)
    close(o16, torch.relu(base + bias), 1e-2, 1e-2, "relu")
    # aux add f32 (in place accumulate) with alpha
    acc = rnd(M, N, seed=5)
    ref = acc + 0.5 * base
    L.gemm(A, B, alpha=0.5, aux=acc, aux_kind=L.AUX_ADD_F32, out_f32=acc)
    close(acc, ref, 1e-4, 1e-3, "add_f32")
    # aux add bf16
    xb = bf(rnd(M, N, seed=6)).to(BF16)
    L.gemm(A, B, aux=xb, aux_kind=L.AUX_ADD_BF16, out_bf16=o16)
    close(o16, base + xb.float(), 1e-2, 1e-2, "add_bf16")
    # dgelu
    hp = bf(rnd(M, N, seed=7)).to(BF16)
    x = hp.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    L.gemm(A, B, aux=hp, aux_kind=L.AUX_MUL_DGELU_BF16, out_bf16=o16)
    close(o16, base * x.grad, 1e-2, 1e-2, "dgelu")
    # gelu with saved derivative + plain bf16 multiply (training path)
    dsave = torch.empty(M, N, dtype=BF16, device=DEV)
    L.gemm(A, B, bias=bias, act=L.ACT_GELU_GRAD, out_bf16=o16, out_pre=dsave)
    pz = (base + bias).requires_grad_(True)
    gz = F.gelu(pz)
    gz.sum().backward()
    close(o16, gz, 1e-2, 1e-2, "gelu (grad variant)")
    close(dsave, pz.grad, 1e-2, 1e-2, "saved gelu'")
    L.gemm(A, B, aux=hp, aux_kind=L.AUX_MUL_BF16, out_bf16=o16)
    close(o16, base * hp.float(), 1e-2, 1e-2, "mul_bf16")
    # positive mask with alpha
    L.gemm(A, B, alpha=1.25, aux=hp, aux_kind=L.AUX_MUL_POS_BF16, out_bf16=o16)
    close(o16, 1.25 * base * (hp.float() > 0), 1e-2, 1e-2, "mul_pos")


def test_gemm_splitk_accumulates_and_batched(L):
    M, N, K = 192, 320, 64 * 37
    A, B = bf(rnd(M, K, seed=1)).to(BF16), bf(rnd(N, K, seed=2, scale=0.1)).to(BF16)
    out = torch.ones(M, N, dtype=F32, device=DEV)
    L.gemm(A, B, out_f32=out, splitk=8)
    close(out, 1.0 + A.float() @ B.float().t(), 1e-4, 2e-3, "splitk atomic")
    ws = torch.empty(1 << 20, dtype=F32, device=DEV)
    out = torch.ones(M, N + 2, dtype=F32, device=DEV)
    L.gemm(A, B, out_f32=out, splitk=8, ws=ws, N=N - 2)  # N not a multiple of 4, workspace path (deterministic fold)
    close(out[:, :N - 2], 1.0 + (A.float() @ B.float().t())[:, :N - 2], 1e-4, 2e-3, "splitk ws")
    assert (out[:, N - 2:] == 1.0).all()
    out2 = torch.ones(M, N + 2, dtype=F32, device=DEV)
    L.gemm(A, B, out_f32=out2, splitk=8, ws=ws, N=N - 2)
    assert torch.equal(out, out2), "workspace split-K must be bit-reproducible"
    # strided batch with per-batch column offset in the output (position-table gradient layout)
    nb, Mb, Nb, Kb = 3, 100, 64, 128
    A3, B3 = bf(rnd(nb, Mb, Kb, seed=4)).to(BF16), bf(rnd(nb, Nb, Kb, seed=5)).to(BF16)
    out = torch.zeros(Mb, nb * Nb + 64, dtype=F32, device=DEV)
    o3 = torch.as_strided(out, (nb, Mb, Nb), (Nb, out.shape[1], 1), 64)
    L.gemm(A3, B3, out_f32=o3, splitk=2, ws=torch.empty(1 << 18, dtype=F32, device=DEV))
    ref = torch.einsum("bmk,bnk->bmn", A3.float(), B3.float())
    for b in range(nb):
        close(out[:, 64 + b * Nb: 64 + (b + 1) * Nb], ref[b], 1e-4, 1e-3, f"batched {b}")
    assert (out[:, :64] == 0).all()
    # few rows against a very long K with a workspace: 8-phase 256 x 256 tiles, K in slices that fill the chip (the prediction
    # head's backward [~700 x 1536 x 128128]); ragged last row / column tile, accumulation into a pre-filled output, reproducible
    for M, N, K in ((691, 1536, 64 * 260), (700, 1100, 64 * 130)):
        A, B = bf(rnd(M, K, seed=7, scale=0.1)).to(BF16), bf(rnd(N, K, seed=8, scale=0.1)).to(BF16)
        ws = torch.empty(24 << 20, dtype=F32, device=DEV)
        out = torch.ones(M, N, dtype=F32, device=DEV)
        L.gemm(A, B, out_f32=out, splitk=4, ws=ws)
        ref = 1.0 + A.float() @ B.float().t()
        close(out, ref, 1e-4, 2e-3 * ref.abs().max().item(), "8-phase split-K")
        out2 = torch.ones(M, N, dtype=F32, device=DEV)
        L.gemm(A, B, out_f32=out2, splitk=4, ws=ws)
        assert torch.equal(out, out2)


@pytest.mark.parametrize("N,H,A,nad", [(8512, 1536, 192, 2), (333, 128, 16, 3), (1000, 200, 100, 1), (2100, 1536, 256, 16), (70, 64, 64, 2)])
def test_adapter_bwd_dw(L, N, H, A, nad):
    """dWu += dy^T z, dWd += dz^T x, dbd += colsum(dz) for a group of adapters in one launch (model/adapter.py:38-42
    autograd): bench shape, a padded bottleneck (A = 16 -> Ap = 64; A = 100 -> 128), a ragged last 64-column tile (H = 200),
    the widest bottleneck with the largest group, fewer rows than a row tile; the first adapter has TWO segments (an adapter
    executed twice); strided wide operand; accumulation into pre-filled outputs; bit-reproducible; NULL outputs."""
    Ap = (A + 63) // 64 * 64

    def seg(seed, folded):
        # folded: dy and dz are column slices of ONE wider buffer ([dy | dz | pad], the folded backward operand);
        # otherwise dy / x are slices of their own padded buffers -- row strides differ from adapter to adapter
        if folded:
            buf = torch.zeros(N, H + Ap + 64, dtype=BF16, device=DEV)
            dy, dz = buf[:, :H], buf[:, H:H + Ap]
        else:
            dy = torch.zeros(N, H + 64, dtype=BF16, device=DEV)[:, :H]
            dz = torch.zeros(N, Ap, dtype=BF16, device=DEV)
        dy.copy_(bf(rnd(N, H, seed=seed)))
        dz[:, :A] = bf(rnd(N, A, seed=seed + 3, scale=0.1))
        xf = torch.zeros(N, H + 64, dtype=BF16, device=DEV)
        xf[:, :H] = bf(rnd(N, H, seed=seed + 1))
        z = torch.zeros(N, Ap, dtype=BF16, device=DEV); z[:, :A] = torch.relu(bf(rnd(N, A, seed=seed + 2)))
        return dy, z, dz, xf[:, :H]

    segs = [[seg(100 * o + 10 * k, o % 2 == 1) for k in range(2 if o == 0 else 1)] for o in range(nad)]
    outs = []
    for _ in range(2):
        grp = [(segs[o], torch.full((H, A), 0.25, dtype=F32, device=DEV), torch.full((A, H), -0.5, dtype=F32, device=DEV),
                torch.full((A,), 1.0, dtype=F32, device=DEV)) for o in range(nad)]
        L.adapter_bwd_dw(grp, A=A)
        outs.append(grp)
    tol = dict(rtol=2e-3, atol=2e-3 * math.sqrt(2 * N))
    for o in (0, nad - 1):
        sg, dWu, dWd, dbd = outs[0][o]
        close(dWu, 0.25 + sum(dy.float().t() @ z[:, :A].float() for dy, z, dz, x in sg), name="dWu", **tol)
        close(dWd, -0.5 + sum(dz[:, :A].float().t() @ x.float() for dy, z, dz, x in sg), name="dWd", **tol)
        close(dbd, 1.0 + sum(dz[:, :A].float().sum(0) for dy, z, dz, x in sg), name="dbd", **tol)
    for ga, gb in zip(outs[0], outs[1]):
        for a, b in zip(ga[1:], gb[1:]):
            assert torch.equal(a, b)
    only = torch.zeros(A, H, dtype=F32, device=DEV)  # any output may be NULL
    L.adapter_bwd_dw([(segs[0], None, only, None)], A=A)
    assert torch.equal(only, outs[0][0][2] + 0.5) or torch.allclose(only, outs[0][0][2] + 0.5, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("K,M,N", [(8512, 1536, 192), (333, 192, 1536), (64, 16, 128), (1000, 128, 16)])
def test_gemm_tn_accumulate(L, K, M, N):
    """C += A^T B with row-major operands (contraction over rows): the dW form, no transposed copies."""
    Mp, Np = (M + 63) // 64 * 64, (N + 63) // 64 * 64
    A = torch.zeros(K, Mp, dtype=BF16, device=DEV); A[:, :M] = bf(rnd(K, M, seed=1))
    B = torch.zeros(K, Np, dtype=BF16, device=DEV); B[:, :N] = bf(rnd(K, N, seed=2, scale=0.1))
    out = torch.ones(M, N, dtype=F32, device=DEV)
    ws = torch.empty(16 << 20, dtype=F32, device=DEV)
    L.gemm_tn_acc(A, B, out, ws, M=M, N=N, splitk=8)
    ref = 1.0 + A[:, :M].float().t() @ B[:, :N].float()
    close(out, ref, 1e-4, 2e-3 * max(1.0, K / 1000), "gemm_tn")
    out2 = torch.ones(M, N, dtype=F32, device=DEV)
    L.gemm_tn_acc(A, B, out2, ws, M=M, N=N, splitk=8)
    assert torch.equal(out, out2)


def test_gemm_strided_views(L):
    """operands / outputs that are column slices of wider buffers (QKV packing)."""
    M, K = 150, 128
    X = bf(rnd(M, 3 * K, seed=1)).to(BF16)
    W = bf(rnd(3 * 64, K, seed=2, scale=0.1)).to(BF16)
    out = torch.zeros(M, 3 * 64, dtype=BF16, device=DEV)
    L.gemm(X[:, K:2 * K], W[64:], out_bf16=out[:, 64:])
    close(out[:, 64:], X[:, K:2 * K].float() @ W[64:].float().t(), 1e-2, 1e-2, "views")
    assert (out[:, :64] == 0).all()


# ------------------------------------------------------------------------------------------------ LayerNorm family
@pytest.mark.parametrize("H", [128, 768, 1536])
def test_ln_fwd_bwd(L, H):
    N, eps = 203, 1e-7
    y, r = rnd(N, H, seed=1), rnd(N, H, seed=2)
    g, b = 1 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    rowmask = (torch.arange(N, device=DEV) % 5 != 0).to(torch.int32)
    t = torch.empty(N, H, device=DEV); st = torch.empty(N, 2, device=DEV)
    ob = torch.empty(N, H, dtype=BF16, device=DEV); of = torch.empty(N, H, device=DEV)
    L.ln_fwd(y=y, r_plain=r, gamma=g, beta=b, eps=eps, rowmask=rowmask, out_t=t, out_stats=st, out_bf16=ob, out_f32=of,
             N=N, H=H)
    tt = (y + r).requires_grad_(True)
    ref = F.layer_norm(tt, (H,), g, b, eps) * rowmask[:, None]
    close(t, y + r, 0, 1e-6, "t")
    close(of, ref, 1e-5, 1e-5, "ln f32")
    close(ob, ref, 1e-2, 1e-2, "ln bf16")
    close(st[:, 0], (y + r).mean(1), 1e-5, 1e-6, "mean")
    # normalised-form residual: second LN consuming the first as residual
    g2, b2 = 1 + 0.1 * rnd(H, seed=5), 0.1 * rnd(H, seed=6)
    y2 = rnd(N, H, seed=7)
    t2 = torch.empty(N, H, device=DEV); st2 = torch.empty(N, 2, device=DEV); of2 = torch.empty(N, H, device=DEV)
    L.ln_fwd(y=y2, r_norm=(t, st, g, b, rowmask), gamma=g2, beta=b2, eps=eps, out_t=t2, out_stats=st2, out_f32=of2, N=N, H=H)
    close(of2, F.layer_norm(y2 + ref.detach(), (H,), g2, b2, eps), 1e-4, 1e-4, "ln chained")
    # materialize with broadcast add
    S = 7
    pos = rnd(S, H, seed=8)
    mf = torch.empty(N, H, device=DEV); mb = torch.empty(N, H, dtype=BF16, device=DEV)
    L.ln_materialize(t, st, g, b, rowmask=rowmask, add_bcast=pos, S=S, out_f32=mf, out_bf16=mb)
    close(mf, ref.detach() + pos[torch.arange(N, device=DEV) % S], 1e-5, 1e-5, "materialize")
    # backward
    dout = rnd(N, H, seed=9)
    ref.backward(dout)
    dg0, db0 = rnd(H, seed=10), rnd(H, seed=11)
    dg, db = dg0.clone(), db0.clone()
    dt = torch.empty(N, H, device=DEV); dyb = torch.empty(N, H, dtype=BF16, device=DEV)
    dys = torch.ones(H, device=DEV)
    L.ln_bwd(dout, t, st, g, rowmask=rowmask, out_dt=dt, out_dy_bf16=dyb, dgamma=dg, dbeta=db, dysum=dys,
             ws=L.ln_bwd_ws(H, DEV))
    close(dt, tt.grad, 1e-4, 1e-4, "dt")
    close(dys - 1.0, tt.grad.sum(0), 1e-4, 1e-3, "colsum(dy)")
    close(dyb, tt.grad, 1e-2, 1e-2, "dy bf16")
    # dy into the first H columns of a wider operand buffer (the [dy | dz] operand of the folded adapter backward)
    wide = torch.full((N, H + 192), 3.0, dtype=BF16, device=DEV)
    L.ln_bwd(dout, t, st, g, rowmask=rowmask, out_dy_bf16=wide[:, :H], ws=L.ln_bwd_ws(H, DEV))
    assert torch.equal(wide[:, :H], dyb) and (wide[:, H:] == 3.0).all()
    xh = (tt.detach() - tt.detach().mean(1, keepdim=True)) * st[:, 1:2]
    close(dg - dg0, (dout * rowmask[:, None] * xh).sum(0), 1e-4, 1e-3, "dgamma")
    close(db - db0, (dout * rowmask[:, None]).sum(0), 1e-4, 1e-3, "dbeta")


def test_ln_dropout_consistency(L):
    """the dropout mask of ln_fwd is regenerated bit-identically by ln_bwd (same seed), rate ~ p, scale 1/(1-p)."""
    N, H, p, seed = 256, 256, 0.1, 1234567
    y = torch.ones(N, H, device=DEV)
    g, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    t = torch.empty(N, H, device=DEV); st = torch.empty(N, 2, device=DEV)
    L.ln_fwd(y=y, p_drop=p, seed=seed, gamma=g, beta=b, eps=1e-7, out_t=t, out_stats=st, N=N, H=H)
    keep = t != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    close(t[keep], torch.full_like(t[keep], 1 / (1 - p)), 1e-6, 1e-6, "scale")
    # backward: grad wrt y must carry the same mask
    dout = rnd(N, H, seed=1)
    dt = torch.empty(N, H, device=DEV); dy = torch.empty(N, H, device=DEV)
    L.ln_bwd(dout, t, st, g, p_drop=p, seed=seed, out_dt=dt, out_dy_f32=dy, ws=L.ln_bwd_ws(H, DEV))
    close(dy, dt * keep / (1 - p), 1e-6, 1e-7, "dy mask")
    # a different seed gives a different mask
    t2 = torch.empty(N, H, device=DEV)
    L.ln_fwd(y=y, p_drop=p, seed=seed + 1, gamma=g, beta=b, eps=1e-7, out_t=t2, out_stats=st, N=N, H=H)
    assert ((t2 != 0) != keep).float().mean().item() > 0.05


# ------------------------------------------------------------------------------------------------ misc row ops
def test_embed_gather(L):
    B, T, Lt, H, V = 3, 4, 9, 128, 50
    ids = torch.randint(0, V, (B, Lt), device=DEV)
    E, vp = rnd(V, H, seed=1), rnd(B * T, H, seed=2)
    out = torch.empty(B * (T + Lt), H, device=DEV)
    L.embed_gather(ids, E, vp, T, out)
    ref = torch.cat([vp.view(B, T, H), E[ids]], 1).view(-1, H)
    assert torch.equal(out, ref)
    out2 = torch.empty(B * Lt, H, device=DEV)
    L.embed_gather(ids, E, None, 0, out2)
    assert torch.equal(out2, E[ids].view(-1, H))


def test_im2col_col2im(L):
    B, S, H = 3, 11, 128
    x = bf(rnd(B * S, H, seed=1))
    col = torch.empty(B * S, 3 * H, dtype=BF16, device=DEV)
    L.im2col3(x.to(BF16), col, B, S, H)
    xp = F.pad(x.view(B, S, H), (0, 0, 1, 1))
    ref = torch.cat([xp[:, 0:S], xp[:, 1:S + 1], xp[:, 2:S + 2]], 2).reshape(B * S, 3 * H)
    assert torch.equal(col.float(), ref)
    # conv1d equivalence of the (H_out, k*H + c) weight layout
    w = bf(rnd(H, H, 3, seed=2, scale=0.05))
    W2 = w.permute(0, 2, 1).reshape(H, 3 * H)
    y = torch.empty(B * S, H, device=DEV)
    L.gemm(col, W2.to(BF16).contiguous(), out_f32=y)
    refc = F.conv1d(x.view(B, S, H).permute(0, 2, 1), w, padding=1).permute(0, 2, 1).reshape(B * S, H)
    close(y, refc, 1e-3, 1e-3, "conv as gemm")
    # col2im = adjoint of im2col
    dcol = rnd(B * S, 3 * H, seed=3)
    dx0 = rnd(B * S, H, seed=4)
    dx = dx0.clone()
    L.col2im3(dcol, dx, B, S, H, 1)
    xr = x.clone().requires_grad_(True)
    xpr = F.pad(xr.view(B, S, H), (0, 0, 1, 1))
    (torch.cat([xpr[:, 0:S], xpr[:, 1:S + 1], xpr[:, 2:S + 2]], 2).reshape(B * S, 3 * H) * dcol).sum().backward()
    close(dx - dx0, xr.grad, 1e-5, 1e-5, "col2im")


def test_dropout_gelu(L):
    n = 5000
    c = rnd(n, seed=1)
    y = torch.empty(n, device=DEV)
    L.dropout_gelu_fwd(c, 0.0, 0, y)
    close(y, F.gelu(c), 1e-5, 1e-6, "gelu")
    cr = c.clone().requires_grad_(True)
    dy = rnd(n, seed=2)
    F.gelu(cr).backward(dy)
    ob = torch.empty(n, dtype=BF16, device=DEV); of = torch.empty(n, device=DEV)
    L.dropout_gelu_bwd(dy, c, 0.0, 0, out_bf16=ob, out_f32=of)
    close(of, cr.grad, 1e-4, 1e-5, "dgelu")
    # with dropout: forward/backward masks agree
    p, seed = 0.25, 99
    L.dropout_gelu_fwd(torch.full_like(c, 10.0), p, seed, y)
    keep = y > 1.0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.03
    L.dropout_gelu_bwd(torch.ones_like(c), torch.full_like(c, 10.0), p, seed, out_f32=of)
    assert ((of > 0.5) == keep).all()


def test_dropout_elementwise(L):
    n, p, seed = 40000, 0.1, 7
    x = torch.ones(n, device=DEV)
    of = torch.empty(n, device=DEV); ob = torch.empty(n, dtype=BF16, device=DEV)
    L.dropout_f32(x, p, seed, out_f32=of, out_bf16=ob)
    assert abs((of != 0).float().mean().item() - 0.9) < 0.01
    assert ((ob.float() != 0) == (of != 0)).all()
    xb = torch.ones(n, dtype=BF16, device=DEV)
    L.dropout_bf16_(xb, p, seed)
    assert ((xb.float() != 0) == (of != 0)).all()  # same (seed, index) -> same decision in every kernel


def test_transpose_colsum(L):
    R, Cc = 333, 200
    x = rnd(R, Cc, seed=1)
    Rp = 384
    out = torch.full((Cc, Rp), 5.0, dtype=BF16, device=DEV)
    L.transpose_to_bf16(x, out)
    close(out[:, :R], x.t(), 1e-2, 1e-2, "transpose f32")
    assert (out[:, R:] == 0).all()
    xb = bf(x).to(BF16)
    L.transpose_to_bf16(xb[:, :64], out[:64], cols=64)
    assert torch.equal(out[:64, :R], xb[:, :64].t())
    acc0 = rnd(Cc, seed=2)
    acc = acc0.clone()
    L.colsum(x, acc, L.colsum_ws(Cc, DEV))
    close(acc - acc0, x.sum(0), 1e-4, 1e-3, "colsum f32")
    acc = acc0.clone()
    L.colsum(xb, acc, L.colsum_ws(Cc, DEV), cols=100)
    close((acc - acc0)[:100], xb.float().sum(0)[:100], 1e-4, 1e-3, "colsum bf16")
    assert torch.equal(acc[100:], acc0[100:])


def test_transpose_batched(L):
    rows, cols, cnt = 192, 1536, 5
    flat = rnd(7 + cnt * (rows * cols + 24), seed=3).to(BF16)
    src_off = torch.tensor([8 + i * (rows * cols + 24) for i in range(cnt)], dtype=torch.int64, device=DEV)
    dst = torch.zeros(cnt, cols, rows, dtype=BF16, device=DEV)
    dst_off = torch.tensor([i * rows * cols for i in range(cnt)], dtype=torch.int64, device=DEV)
    L.transpose_batched_bf16(flat, src_off, dst, dst_off, rows, cols)
    for i in range(cnt):
        o = int(src_off[i])
        assert torch.equal(dst[i], flat[o:o + rows * cols].view(rows, cols).t())
    # ragged shape (not a multiple of the 64x64 tile)
    rows, cols = 70, 33
    flat = rnd(2 * rows * cols, seed=4).to(BF16)
    off = torch.tensor([0, rows * cols], dtype=torch.int64, device=DEV)
    dst = torch.zeros(2, cols, rows, dtype=BF16, device=DEV)
    L.transpose_batched_bf16(flat, off, dst, off, rows, cols)
    assert torch.equal(dst[1], flat[rows * cols:].view(rows, cols).t())


def test_head_transpose(L):
    B, S, nh = 2, 70, 3
    Sp = 128
    x = bf(rnd(B * S, 3 * nh * 64, seed=1)).to(BF16)
    v = x[:, 2 * nh * 64:]
    for hm in (False, True):
        vt = torch.full((B * nh * 64 * Sp,), 3.0, dtype=BF16, device=DEV)
        L.head_transpose(v, vt, B, S, Sp, nh, head_major=hm)
        ref = v.float().view(B, S, nh, 64).permute(0, 2, 3, 1)  # [B,nh,64,S]
        got = vt.view(nh, 64, B, Sp).permute(2, 0, 1, 3) if hm else vt.view(B, nh, 64, Sp)
        assert torch.equal(got[..., :S].float(), ref)
        assert (got[..., S:] == 0).all()


def test_attn_bwd_prep_equals_the_separate_launches(L):
    """fbl_attn_bwd_prep = four head transposes + rowdot in one launch: bit-identical to the stand-alone entry points"""
    B, S, nh, span2 = 3, 150, 4, 512
    H, Sp = nh * 64, 192
    qkv = bf(rnd(B * S, 3 * H, seed=1)).to(BF16)
    pqk = bf(rnd(span2, 2 * H, seed=2)).to(BF16)
    dO, O = bf(rnd(B * S, H, seed=3)).to(BF16), bf(rnd(B * S, H, seed=4)).to(BF16)
    q, k, pq, pk = qkv[:, :H], qkv[:, H:2 * H], pqk[:, :H], pqk[:, H:]
    mk = lambda *shape: torch.full(shape, 7.0, dtype=BF16, device=DEV)
    QT, KT, PQT, PKT = mk(nh, 64, B, Sp), mk(nh, 64, B, Sp), mk(nh, 64, span2), mk(nh, 64, span2)
    Dv = torch.full((B, nh, S), 7.0, device=DEV)
    L.attn_bwd_prep(q, k, pq, pk, dO, O, QT, KT, PQT, PKT, Dv, B, S, Sp, nh, span2)
    QT2, KT2, PQT2, PKT2 = mk(nh, 64, B, Sp), mk(nh, 64, B, Sp), mk(nh, 64, span2), mk(nh, 64, span2)
    Dv2 = torch.empty(B, nh, S, device=DEV)
    L.head_transpose(q, QT2, B, S, Sp, nh, head_major=True)
    L.head_transpose(k, KT2, B, S, Sp, nh, head_major=True)
    L.head_transpose(pq, PQT2, 1, span2, span2, nh, head_major=False)
    L.head_transpose(pk, PKT2, 1, span2, span2, nh, head_major=False)
    L.attn_rowdot(dO, O, Dv2, B, S, nh)
    for a, b_, n in ((QT, QT2, "QT"), (KT, KT2, "KT"), (PQT, PQT2, "PQT"), (PKT, PKT2, "PKT"), (Dv, Dv2, "D")):
        assert torch.equal(a, b_), n
    assert (QT[:, :, :, S:] == 0).all()  # positions beyond S are zero-padded
    # optional outputs: only what is asked for is written; the index-expanded tables of the fused key-major pass
    from frozenbilm_amd.model.relpos import rel_index_vector
    relidx = torch.from_numpy(rel_index_vector(S, 256, 512, 256).copy()).to(DEV)
    KT3, PKT3, Dv3 = mk(nh, 64, B, Sp), mk(nh, 64, span2), torch.empty(B, nh, S, device=DEV)
    PQX, PKX = mk(nh, 2 * Sp, 64), mk(nh, 2 * Sp, 64)
    L.attn_bwd_prep(q, k, pq, pk, dO, O, None, KT3, None, PKT3, Dv3, B, S, Sp, nh, span2, relidx=relidx, PQX=PQX, PKX=PKX)
    assert torch.equal(KT3, KT2) and torch.equal(PKT3, PKT2) and torch.equal(Dv3, Dv2)
    t = torch.arange(2 * Sp, device=DEV)
    rows = relidx[(t - Sp + S - 1).clamp(0, 2 * S - 2)].long()  # table row of delta = t - Sp
    for X, tab, n in ((PQX, pq, "PQX"), (PKX, pk, "PKX")):
        ref = tab[rows].view(2 * Sp, nh, 64).permute(1, 0, 2)  # [nh, 2 Sp, 64]
        assert torch.equal(X.view(nh, 2 * Sp, 64), ref.contiguous()), n


def test_cross_entropy(L):
    N, V = 37, 1003
    Vp = 1024
    logits = torch.zeros(N, Vp, device=DEV)
    logits[:, :V] = rnd(N, V, seed=1, scale=3.0)
    labels = torch.randint(0, V, (N,), device=DEV)
    labels[::3] = -100
    lse = torch.empty(N, device=DEV); acc = torch.zeros(2, device=DEV)
    L.ce_fwd(logits, labels, V, lse, acc)
    x = logits[:, :V].clone().requires_grad_(True)
    ref = F.cross_entropy(x, labels, ignore_index=-100)
    assert abs((acc[0] / acc[1]).item() - ref.item()) < 1e-4
    ref.backward()
    rows = torch.nonzero(labels != -100).view(-1).to(torch.int32)
    d = torch.empty(rows.numel(), Vp, dtype=BF16, device=DEV)
    L.ce_bwd_rows(logits, labels, rows, V, Vp, lse, acc, 1.0, d)
    close(d[:, :V], x.grad[rows.long()], 2e-2, 1e-5, "dlogits")
    assert (d[:, V:] == 0).all()
    # gather / scatter rows
    src = bf(rnd(N, 64, seed=2)).to(BF16)
    g = torch.empty(rows.numel(), 64, dtype=BF16, device=DEV)
    L.gather_rows_bf16(src, rows, g)
    assert torch.equal(g, src[rows.long()])
    dst = torch.ones(N, 64, device=DEV)
    upd = rnd(rows.numel(), 64, seed=3)
    L.scatter_rows_f32(upd, rows, dst)
    ref = torch.ones(N, 64, device=DEV)
    ref[rows.long()] += upd
    close(dst, ref, 0, 1e-6, "scatter")


def test_adam_and_sumsq(L):
    n = 10007
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    ss = torch.zeros(1, device=DEV)
    L.sumsq(g, ss)
    assert abs(ss.item() - (g.double() ** 2).sum().item()) / ss.item() < 1e-5
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=3e-4, betas=(0.9, 0.95), eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pt.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pt], 0.1)
        opt.step()
        ss.zero_()
        L.sumsq(g, ss)
        L.adam_flat(p, g, m, v, 3e-4, 0.9, 0.95, 1e-8, 0.0, step, sumsq_t=ss, max_norm=0.1)
    close(p, pt.detach(), 1e-5, 1e-6, "adam+clip")


@pytest.mark.parametrize("H,ds,N", [(128, 8, 1000), (1536, 8, 700)])
def test_adapter_module_gate_matched(L, H, ds, N):
    """frozenbilm_amd.model.Adapter (fwd + bwd through the C ABI) vs torch autograd fed the SAME bf16-rounded
    operands, so both sides open exactly the same ReLU gates (ref: model/adapter.py:33-45)."""
    from frozenbilm_amd.model import Adapter

    torch.manual_seed(0)
    ad = Adapter(ds, H, dropout=0.1).to(DEV).eval()
    with torch.no_grad():
        for p_ in ad.parameters():
            p_.copy_(bf(rnd(*p_.shape, seed=p_.numel() % 97, scale=0.05)))
        ad.down.bias.copy_(rnd(H // ds, seed=5, scale=0.05))
        ad.up.bias.copy_(rnd(H, seed=6, scale=0.05))
    x = bf(rnd(N, H, seed=1)).requires_grad_(True)
    y = ad(x)
    gy = bf(rnd(N, H, seed=2))
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    wd, bd = ad.down.weight.detach().clone().requires_grad_(True), ad.down.bias.detach().clone().requires_grad_(True)
    wu, bu = ad.up.weight.detach().clone().requires_grad_(True), ad.up.bias.detach().clone().requires_grad_(True)
    z = bf(torch.relu(F.linear(xr, wd, bd)))  # the kernel stores z in bf16
    zr = torch.relu(F.linear(xr, wd, bd))
    zz = zr + (z - zr).detach()
    yr = xr + F.linear(zz, wu, bu)
    yr.backward(gy)
    close(y, yr, 1e-3, 6e-3, "adapter y")  # z is rounded to bf16: 1-ulp differences at rounding boundaries
    sc = lambda t: t.abs().max().item()
    close(x.grad, xr.grad, 2e-2, 1e-2 * sc(xr.grad), "dx")
    close(ad.up.weight.grad, wu.grad, 2e-2, 1e-2 * sc(wu.grad), "dWu")
    close(ad.up.bias.grad, bu.grad, 1e-2, 1e-2 * sc(bu.grad), "dbu")
    close(ad.down.weight.grad, wd.grad, 2e-2, 2e-2 * sc(wd.grad), "dWd")
    close(ad.down.bias.grad, bd.grad, 2e-2, 2e-2 * sc(bd.grad), "dbd")
    # train mode: dropout on z is live and consistent between forward and backward
    ad.train()
    y2 = ad(x.detach())
    assert (y2 - y.detach()).abs().max().item() > 0


# ------------------------------------------------------------------------------------------------ attention
def _attn_inputs(B, S, nh, seed, span2=512):
    from frozenbilm_amd.model.relpos import rel_index_vector

    H = nh * 64
    qkv = bf(rnd(B * S, 3 * H, seed=seed, scale=1.0)).to(BF16)
    pqk = bf(rnd(span2, 2 * H, seed=seed + 1, scale=1.0)).to(BF16)
    mask = torch.ones(B, S, dtype=torch.int32, device=DEV)
    mask[0, max(1, S - 5):] = 0
    if B > 1:
        mask[1, 1:3] = 0
    relidx = torch.from_numpy(rel_index_vector(S, 256, 512, 256).copy()).to(DEV)
    return qkv, pqk, mask, relidx, H


def _klen(mask):
    S = mask.shape[1]
    return (mask * torch.arange(1, S + 1, device=mask.device, dtype=torch.int32)).amax(1).to(torch.int32).contiguous()


def _border(klen):
    return torch.argsort(klen, descending=True, stable=True).to(torch.int32).contiguous()


def _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, p_drop=0.0, seed=0, klen=None, border=None, lin=128, save_p=None):
    """save_p: a list that receives (psave, msave) -- the training forward's saved probabilities, allocated NaN-filled so that
    anything the backward reads without the forward having written it shows up"""
    H = nh * 64
    Sp = (S + 63) // 64 * 64
    ctx = torch.zeros(B * S, H, dtype=BF16, device=DEV)
    lse = torch.empty(B, nh, S, device=DEV)
    ps = ms = None
    if save_p is not None:
        ps = torch.full((B, nh, Sp, Sp), float("nan"), dtype=BF16, device=DEV)
        ms = torch.full((B, nh, Sp // 64, S), float("nan"), dtype=torch.float32, device=DEV)
        save_p[:] = [ps, ms]
    L.disent_attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], pqk[:, H:], pqk[:, :H], relidx, mask.view(-1), 1 / math.sqrt(192), ctx,
                      lse, B, S, Sp, nh, pqk.shape[0], p_drop=p_drop, seed=seed, klen=klen, border=border, lin=lin, psave=ps, msave=ms)
    return ctx, lse


# nh * B a multiple of 8 takes the XCD-aware workgroup mapping, everything else the plain one
@pytest.mark.parametrize("B,S,nh", [(1, 16, 1), (2, 37, 2), (2, 64, 1), (3, 129, 2), (2, 266, 3), (1, 512, 2), (4, 150, 2),
                                    (8, 266, 3)])
def test_attention_fwd(L, B, S, nh):
    qkv, pqk, mask, relidx, H = _attn_inputs(B, S, nh, seed=10 + S)
    ctx, lse = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh)
    q, k, v = (heads(qkv[:, i * H:(i + 1) * H].float(), B, S, nh) for i in range(3))
    pq = pqk[:, :H].float().view(-1, nh, 64).permute(1, 0, 2)
    pk = pqk[:, H:].float().view(-1, nh, 64).permute(1, 0, 2)
    ref, rlse = ref_attention(q, k, v, pk, pq, relidx, mask, 1 / math.sqrt(192))
    got = heads(ctx.float(), B, S, nh)
    # scores are O(10) with unit-variance operands: P is bf16-rounded before P.V -> 1e-2 relative
    close(got, ref, 2e-2, 2e-2, f"ctx S={S}")
    valid = mask.bool()[:, None, :].expand(B, nh, S)
    close(lse[valid], rlse[valid], 1e-3, 1e-3, "lse")
    assert torch.isinf(lse[~valid]).all()
    assert (got[~valid[:, :, :, None].expand_as(got)] == 0).all()  # masked query rows are exactly zero
    # skipping the tiles beyond the last valid position (klen) is exact
    mask2 = mask.clone()
    mask2[0, S // 3:] = 0
    c_full, l_full = _run_attn_fwd(L, qkv, pqk, mask2, relidx, B, S, nh)
    c_skip, l_skip = _run_attn_fwd(L, qkv, pqk, mask2, relidx, B, S, nh, klen=_klen(mask2))
    assert torch.equal(c_full, c_skip) and torch.equal(l_full, l_skip)
    # the dispatch order of the samples (longest first) changes nothing
    c_ord, l_ord = _run_attn_fwd(L, qkv, pqk, mask2, relidx, B, S, nh, klen=_klen(mask2), border=_border(_klen(mask2)))
    assert torch.equal(c_full, c_ord) and torch.equal(l_full, l_ord)
    # index-table-free addressing inside the identity band (lin = position_buckets / 2) gathers the very same values
    c_tab, l_tab = _run_attn_fwd(L, qkv, pqk, mask2, relidx, B, S, nh, lin=0)
    assert torch.equal(c_full, c_tab) and torch.equal(l_full, l_tab)


def test_attention_fwd_dropout_rate(L):
    B, S, nh = 1, 128, 1
    qkv, pqk, mask, relidx, H = _attn_inputs(B, S, nh, seed=3)
    mask[:] = 1
    qkv[:, :2 * H] = 0  # uniform attention
    pqk[:] = 0
    qkv[:, 2 * H:] = 1.0  # V = 1 -> ctx = sum of kept probs / (1-p) ~ 1
    ctx, _ = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, p_drop=0.1, seed=5)
    m = ctx.float().mean().item()
    assert abs(m - 1.0) < 0.02, m
    assert ctx.float().std().item() > 1e-3  # not all ones: dropout really dropped something


@pytest.mark.parametrize("saved_p", [False, True, "gt_route", "separate_dk"], ids=["recompute", "saved_p", "gt_route", "separate_dk"])
@pytest.mark.parametrize("B,S,nh", [(1, 16, 1), (2, 37, 2), (2, 130, 2), (2, 266, 2), (3, 266, 1), (2, 512, 1), (4, 200, 2)])
def test_attention_bwd(L, B, S, nh, saved_p):
    """saved_p: kernel A reads the un-normalised probabilities the (training) forward left in HBM (fbl_disent_attn_bwd_dsp)
    instead of recomputing the scores; the forward then runs with the same klen / border as the backward (it writes exactly
    the tile pairs the backward reads -- the buffers start NaN-filled)."""
    from frozenbilm_amd.attn_bwd import disent_attn_bwd

    qkv, pqk, mask, relidx, H = _attn_inputs(B, S, nh, seed=20 + S)
    if B >= 3:  # short samples: whole 64-row steps of G^T beyond klen stay unwritten and must be skipped downstream
        mask[1, 70:] = 0
        mask[2, 33:] = 0
    if S == 512:
        mask[1, 200:] = 0
    qkv = (qkv.float() * 0.5).to(BF16)
    pqk = (pqk.float() * 0.5).to(BF16)
    klen_t = _klen(mask) if S > 100 else None  # exercise both the dense and the tile-skipping paths
    gt_route = saved_p == "gt_route"  # position-table gradients through G^T + split-K GEMMs (rounds 1-5) instead of fbl_attn_pos_grad
    separate_dk = saved_p == "separate_dk"  # saved probabilities, dK by the key-major shear pass instead of inside kernel A
    saved_p = saved_p is True or separate_dk
    saved = [] if saved_p else None
    ctx, lse = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, klen=klen_t, save_p=saved)
    dctx = bf(rnd(B * S, H, seed=5)).to(BF16)
    # reference grads by autograd
    qkvf = qkv.float().requires_grad_(True)
    pqkf = pqk.float().requires_grad_(True)
    q, k, v = (heads(qkvf[:, i * H:(i + 1) * H], B, S, nh) for i in range(3))
    pq = pqkf[:, :H].view(-1, nh, 64).permute(1, 0, 2)
    pk = pqkf[:, H:].view(-1, nh, 64).permute(1, 0, 2)
    ref, _ = ref_attention(q, k, v, pk, pq, relidx, mask, 1 / math.sqrt(192))
    (unheads(ref) * dctx.float()).sum().backward()

    class E:  # minimal engine/run/sv stand-ins
        pass

    eng, run, sv = E(), E(), E()
    eng.H, eng.nh, eng.span2, eng.dev = H, nh, pqk.shape[0], torch.device(DEV)
    eng.relidx = lambda S_: relidx
    import types as _t
    eng.cfg = _t.SimpleNamespace(position_buckets=256, max_rel=512, att_span=256)  # enables the relidx-range / injective-store paths
    run.B, run.S, run.mask_i32, run.p_att = B, S, mask.view(-1), 0.0
    eng.pos_grad_gt = gt_route
    eng.attn_fused_dk = eng.attn_toeplitz_dq = not separate_dk  # "separate_dk": both shear passes of rounds 1-5
    run.klen = klen_t
    run.border = _border(run.klen) if (run.klen is not None and B >= 3) else None  # longest-first dispatch (XCD-aware map at B=4)
    import frozenbilm_amd.attn_bwd as AB
    AB.POISON_GT = True  # unwritten G^T blocks hold NaN: the position-table GEMMs must skip exactly those
    sv.qkv, sv.pqk, sv.ctx, sv.lse, sv.seed_att = qkv, pqk, ctx, lse, 0
    if saved_p:
        sv.psave, sv.msave = saved
    dqkv = torch.zeros(B * S, 3 * H, dtype=BF16, device=DEV)
    dpqk = torch.zeros(pqk.shape[0], 2 * H, dtype=BF16, device=DEV)
    disent_attn_bwd(eng, run, sv, dctx, dqkv, dpqk)
    gq = qkvf.grad
    sc = gq.abs().max().item()
    for name, sl in (("dQ", slice(0, H)), ("dK", slice(H, 2 * H)), ("dV", slice(2 * H, 3 * H))):
        close(dqkv[:, sl], gq[:, sl], 3e-2, 2e-2 * sc, name)
    sp = pqkf.grad.abs().max().item()
    close(dpqk[:, H:], pqkf.grad[:, H:], 3e-2, 2e-2 * sp, "dPK")
    close(dpqk[:, :H], pqkf.grad[:, :H], 3e-2, 2e-2 * sp, "dPQ")


def _attn_bwd_call(L, qkv, pqk, ctx, lse, dctx, mask, relidx, B, S, nh, H, p_att, seed, saved=None):
    from frozenbilm_amd.attn_bwd import disent_attn_bwd
    import types as _t

    class E:
        pass

    eng, run, sv = E(), E(), E()
    eng.H, eng.nh, eng.span2, eng.dev = H, nh, pqk.shape[0], torch.device(DEV)
    eng.relidx = lambda S_: relidx
    eng.cfg = _t.SimpleNamespace(position_buckets=256, max_rel=512, att_span=256)
    run.B, run.S, run.mask_i32, run.p_att = B, S, mask.view(-1), p_att
    run.klen = None
    sv.qkv, sv.pqk, sv.ctx, sv.lse, sv.seed_att = qkv, pqk, ctx, lse, seed
    if saved:
        sv.psave, sv.msave = saved
    dqkv = torch.zeros(B * S, 3 * H, dtype=BF16, device=DEV)
    dpqk = torch.zeros(pqk.shape[0], 2 * H, dtype=BF16, device=DEV)
    disent_attn_bwd(eng, run, sv, dctx, dqkv, dpqk)
    return dqkv, dpqk


@pytest.mark.parametrize("saved_p", [False, True], ids=["recompute", "saved_p"])
def test_attention_dropout_mask_fwd_bwd_agree(L, saved_p):
    """V = I and dO = I expose the dropped-out probability matrix in both directions: forward ctx = drop(P), backward
    dV = drop(P)^T -- the two kernels must regenerate the SAME mask; its statistics must look like iid Bernoulli(1-p)."""
    B, S, nh, p = 2, 64, 2, 0.1
    qkv, pqk, mask, relidx, H = _attn_inputs(B, S, nh, seed=77)
    mask[:] = 1
    qkv = (qkv.float() * 0.3).to(BF16)  # mild scores: every P entry is well above bf16 underflow
    eye = torch.eye(64, device=DEV).repeat(B, nh).to(BF16)  # V[b*S+s, h*64+d] = (s == d)
    qkv[:, 2 * H:] = eye
    saved = [] if saved_p else None
    ctx, lse = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, p_drop=p, seed=1234, save_p=saved)
    dqkv, _ = _attn_bwd_call(L, qkv, pqk, ctx, lse, eye.clone(), mask, relidx, B, S, nh, H, p, 1234, saved=saved)
    Pf = heads(ctx.float(), B, S, nh)                 # [B,nh,S(query),64(key)]
    Pb = heads(dqkv[:, 2 * H:].float(), B, S, nh)     # [B,nh,S(key),64(query)]
    assert torch.equal(Pf == 0, Pb.transpose(-1, -2) == 0)
    close(Pf, Pb.transpose(-1, -2), 2e-2, 1e-4, "drop(P) fwd vs bwd")
    drop = (Pf == 0).float()
    n = drop.numel()
    rate = drop.mean().item()
    assert abs(rate - p) < 4 * math.sqrt(p * (1 - p) / n), rate
    d0 = drop - rate
    for name, a_, b_ in (("key+1", d0[..., :, :-1], d0[..., :, 1:]), ("query+1", d0[..., :-1, :], d0[..., 1:, :]),
                         ("diag", d0[..., :-1, :-1], d0[..., 1:, 1:])):
        corr = (a_ * b_).mean().item() / (p * (1 - p))
        assert abs(corr) < 5 / math.sqrt(a_.numel()), (name, corr)
    # a different seed gives a different mask; the (batch, head) streams differ from each other
    ctx2, _ = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, p_drop=p, seed=1235)
    assert not torch.equal(ctx2 == 0, ctx == 0)
    assert not torch.equal(drop[0, 0], drop[0, 1]) and not torch.equal(drop[0, 0], drop[1, 0])


@pytest.mark.parametrize("saved_p", [False, True], ids=["recompute", "saved_p"])
def test_attention_dropout_bwd_linearity(L, saved_p):
    """multi-tile, ragged: ctx is linear in V for a fixed mask, so <ctx, dO> == <V, dV> iff backward uses the forward mask."""
    B, S, nh, p = 2, 150, 2, 0.1
    qkv, pqk, mask, relidx, H = _attn_inputs(B, S, nh, seed=78)
    qkv = (qkv.float() * 0.5).to(BF16)
    saved = [] if saved_p else None
    ctx, lse = _run_attn_fwd(L, qkv, pqk, mask, relidx, B, S, nh, p_drop=p, seed=99, save_p=saved)
    dctx = bf(rnd(B * S, H, seed=6)).to(BF16)
    dqkv, _ = _attn_bwd_call(L, qkv, pqk, ctx, lse, dctx, mask, relidx, B, S, nh, H, p, 99, saved=list(saved) if saved else None)
    for h in range(nh):
        sl = slice(h * 64, (h + 1) * 64)
        lhs = (ctx[:, sl].float() * dctx[:, sl].float()).sum().item()
        rhs = (qkv[:, 2 * H:][:, sl].float() * dqkv[:, 2 * H:][:, sl].float()).sum().item()
        assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), 10.0), (h, lhs, rhs)
    dq2, _ = _attn_bwd_call(L, qkv, pqk, ctx, lse, dctx, mask, relidx, B, S, nh, H, p, 100, saved=saved)
    if saved_p:
        # the mask travels with the saved probabilities (their sign bits): the backward does not read the seed at all
        assert torch.equal(dq2, dqkv)
    else:
        # wrong seed in backward breaks the identity by far more than the tolerance
        assert (dq2[:, 2 * H:].float() - dqkv[:, 2 * H:].float()).abs().max().item() > 1e-2


# ------------------------------------------------------------------------------------------------ input side
def test_video_stage_matches_reference_dataset(L, golden, tmp_path):
    """packed fp16 clips -> [B,T,F]: bit-exact against what the REFERENCE's dataset class produced (golden G12)"""
    from frozenbilm_amd.datasets import PackedVideoText_Dataset, packed_collate_fn, stage_packed_batch
    from tests.golden.make_goldens import write_feature_fixture

    g = golden("G12_dataset", raw=True)
    csv_path, feats = write_feature_fixture(str(tmp_path))
    ds = PackedVideoText_Dataset(csv_path, feats, max_feats=10, features_dim=16)
    batch = stage_packed_batch(packed_collate_fn([ds[i] for i in range(len(ds))]), 10, DEV)
    assert torch.equal(batch["video"].cpu(), torch.from_numpy(g["video"]))
    assert torch.equal(batch["video_len"].cpu(), torch.from_numpy(g["video_len"]))
    from frozenbilm_amd.util.misc import get_mask

    assert torch.equal(batch["video_mask"].cpu(), get_mask(torch.from_numpy(g["video_len"]), 10))
    # bench-shaped case: 1024-wide features, many clips
    gen = torch.Generator().manual_seed(5)
    n = torch.randint(0, 40, (64,), generator=gen).to(torch.int32)
    feats = torch.randn(int(n.sum()), 1024, generator=gen).half()
    off = torch.zeros(64, dtype=torch.int64); off[1:] = torch.cumsum(n[:-1].long(), 0)
    video, vlen, vmask = L.video_stage_f16(feats.to(DEV), off.to(DEV), n.to(DEV), 10)
    for b in range(64):
        nb = int(n[b]); clip = feats[off[b]: off[b] + nb].float()
        want = torch.zeros(10, 1024)
        if nb > 10:
            want = clip[(torch.arange(10) * nb) // 10]
        else:
            want[:nb] = clip
        assert torch.equal(video[b].cpu(), want) and int(vlen[b]) == min(nb, 10)


def test_mask_tokens_device(L):
    from frozenbilm_amd.util.misc import mask_tokens_device
    from tests.downstream_fixtures import StubTokenizer

    tok = StubTokenizer(30000)
    gen = torch.Generator().manual_seed(1)
    B, Lt = 256, 128
    ids = torch.randint(5, 30000, (B, Lt), generator=gen)
    ids[:, 0] = tok.cls_token_id
    tl = torch.randint(8, Lt, (B,), generator=gen)
    for b in range(B):
        ids[b, tl[b]] = tok.sep_token_id
        ids[b, tl[b] + 1:] = tok.pad_token_id
    orig = ids.clone()
    x = ids.to(DEV)
    out, labels = mask_tokens_device(x, tok, 0.15, seed=42)
    assert out.data_ptr() == x.data_ptr()  # in place, like the reference
    out, labels = out.cpu(), labels.cpu()
    special = (orig == tok.pad_token_id) | (orig == tok.cls_token_id) | (orig == tok.sep_token_id)
    sel = labels != -100
    assert not (sel & special).any() and torch.equal(labels[sel], orig[sel]) and torch.equal(out[~sel], orig[~sel])
    n_ok = int((~special).sum()); n_sel = int(sel.sum())
    assert abs(n_sel / n_ok - 0.15) < 4 * (0.15 * 0.85 / n_ok) ** 0.5
    masked = sel & (out == tok.mask_token_id)
    kept = sel & (out == orig)
    rand = sel & ~masked & ~kept
    assert abs(int(masked.sum()) / n_sel - 0.8) < 0.02 and abs(int(kept.sum()) / n_sel - 0.1) < 0.015
    assert abs(int(rand.sum()) / n_sel - 0.1) < 0.015 and int(out[rand].min()) >= 0 and int(out[rand].max()) < 30000
    # deterministic in the seed, different across seeds
    y = orig.to(DEV); o2, l2 = mask_tokens_device(y, tok, 0.15, seed=42)
    assert torch.equal(o2.cpu(), out) and torch.equal(l2.cpu(), labels)
    z = orig.to(DEV); o3, l3 = mask_tokens_device(z, tok, 0.15, seed=43)
    assert not torch.equal(l3.cpu(), labels)


@pytest.mark.parametrize("M,N1,A,K", [(300, 128, 16, 128), (4100, 1536, 192, 256), (8512, 1536, 192, 1536), (1000, 768, 96, 3072),
                                      (2500, 192, 24, 128), (4100, 1152, 144, 256)])
# small tiles; 8-phase 256x256 tiles (7th tile column = the bottleneck); the Wo shape of the step; a mid-size one;
# N1 = 192 (a multiple of 64 only); N1 = 1152 = 4.5 x 256: big-tile shape whose boundary would cut a wave's two 32-column
# ranges -> must run on the narrow tiles
def test_dense_adapter_down_merged(L, M, N1, A, K):
    """fbl_dense_adapter_down_fwd: y = x.W^T + b and z = dropout(relu(y.Wd^T + bd)) from ONE GEMM against [W ; Wd.W]."""
    x = bf(rnd(M, K, seed=1)).to(BF16)
    W = bf(rnd(N1, K, seed=2, scale=0.05)).to(BF16)
    b = rnd(N1, seed=3, scale=0.1)
    Wd = bf(rnd(A, N1, seed=4, scale=0.05)).to(BF16)
    bd = rnd(A, seed=5, scale=0.1)
    # composed rows / bias the way the engine builds them (fbl_gemm_bf16_nt on W^T)
    WT = W.t().contiguous()
    Wm = torch.zeros(N1 + A, K, dtype=BF16, device=DEV)
    Wm[:N1] = W
    L.gemm(Wd, WT, out_bf16=Wm[N1:])
    bm = torch.cat([b, Wd.float() @ b + bd]).contiguous()
    y32 = torch.empty(M, N1, dtype=F32, device=DEV)
    y16 = torch.empty(M, N1, dtype=BF16, device=DEV)
    z = torch.full((M, A), 7.0, dtype=BF16, device=DEV)
    L.dense_adapter_down_fwd(x, Wm, bm, N1, z, y_f32=y32, y_bf16=y16)
    yref = x.float() @ W.float().t() + b
    close(y32, yref, 1e-4, 2e-3, "y f32")
    close(y16, yref, 1e-2, 1e-2, "y bf16")
    zref = torch.relu(yref @ Wd.float().t() + bd)
    close(z, zref, 2e-2, 2e-2 * max(1.0, zref.abs().max().item()), "z")
    # dropout: same keys as fbl_dropout_bf16 on the [M, A] tensor, kept values scaled by 1/(1-p)
    p, seed = 0.25, 1234
    z2 = torch.empty(M, A, dtype=BF16, device=DEV)
    L.dense_adapter_down_fwd(x, Wm, bm, N1, z2, y_f32=y32, y_bf16=y16, p_drop=p, seed=seed)
    ones = torch.ones(M, A, dtype=BF16, device=DEV)
    L.dropout_bf16_(ones, p, seed)
    keep = ones.float() > 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    assert (z2.float()[~keep] == 0).all()
    if N1 == 128:  # the contract: a segment boundary that is not a multiple of 64 is refused, not silently mis-staged
        with pytest.raises(RuntimeError):
            L.dense_adapter_down_fwd(x[:, :K], Wm[: 100 + A], bm[: 100 + A].contiguous(), 100, z, y_f32=y32[:, :100].contiguous())
    close(z2.float()[keep], z.float()[keep] / (1 - p), 2e-2, 1e-2, "kept values")


@pytest.mark.parametrize("M,H,A", [(203, 128, 64), (8512, 1536, 192), (1000, 768, 128), (4100, 1536, 192), (333, 256, 64)])
def test_adapter_up_resid_tail(L, M, H, A):
    """fbl_adapter_up_resid_fwd: t = dropout(x + z.Wu^T + bu) + residual, residual plain or LayerNorm-normalised with a row
    mask; the dropout mask is the one fbl_ln_fwd / fbl_ln_bwd generate from the same seed; fbl_ln_fwd(y = t) then gives the
    same statistics / bf16 operand as the unfused block (up GEMM -> fbl_ln_fwd with dropout and residual)."""
    eps = 1e-7
    z = bf(torch.relu(rnd(M, A, seed=1))).to(BF16)
    Wu = bf(rnd(H, A, seed=2, scale=0.05)).to(BF16)
    bu = rnd(H, seed=3, scale=0.1)
    x = bf(rnd(M, H, seed=4)).to(BF16)
    r = rnd(M, H, seed=5)
    y_ref = x.float() + z.float() @ Wu.float().t() + bu
    # (a) plain residual, no dropout
    t = torch.full((M, H), 9.0, device=DEV)
    L.adapter_up_resid_fwd(z, Wu, bu, x, t, r_plain=r)
    close(t, y_ref + r, 1e-4, 2e-3, "plain residual")
    # (b) normalised residual with row mask: the representation fbl_ln_fwd leaves behind
    g, b = 1 + 0.1 * rnd(H, seed=6), 0.1 * rnd(H, seed=7)
    rowmask = (torch.arange(M, device=DEV) % 7 != 0).to(torch.int32)
    rt = torch.empty(M, H, device=DEV); rst = torch.empty(M, 2, device=DEV); rf = torch.empty(M, H, device=DEV)
    L.ln_fwd(y=r, gamma=g, beta=b, eps=eps, rowmask=rowmask, out_t=rt, out_stats=rst, out_f32=rf, N=M, H=H)
    t2 = torch.empty(M, H, device=DEV)
    L.adapter_up_resid_fwd(z, Wu, bu, x, t2, r_norm=(rt, rst, g, b, rowmask))
    close(t2, y_ref + rf, 1e-4, 2e-3, "normalised residual")
    # (c) dropout: against the unfused block on the same fp32 adapter output
    p, seed = 0.1, 424242
    t3 = torch.empty(M, H, device=DEV)
    L.adapter_up_resid_fwd(z, Wu, bu, x, t3, p_drop=p, seed=seed, r_norm=(rt, rst, g, b, rowmask))
    y32 = torch.empty(M, H, device=DEV)
    L.gemm(z, Wu, bias=bu, aux=x.float().contiguous(), aux_kind=L.AUX_ADD_F32, out_f32=y32)
    g2, b2 = 1 + 0.1 * rnd(H, seed=8), 0.1 * rnd(H, seed=9)
    t_ref = torch.empty(M, H, device=DEV); st_ref = torch.empty(M, 2, device=DEV); ob_ref = torch.empty(M, H, dtype=BF16, device=DEV)
    L.ln_fwd(y=y32, p_drop=p, seed=seed, r_norm=(rt, rst, g, b, rowmask), gamma=g2, beta=b2, eps=eps, out_t=t_ref,
             out_stats=st_ref, out_bf16=ob_ref, N=M, H=H)
    close(t3, t_ref, 1e-4, 2e-3, "dropout + residual vs unfused block")
    dropped = (t3 - rf).abs() < 1e-6  # dropped elements: t == residual
    assert abs(dropped.float().mean().item() - p) < 0.02
    st = torch.empty(M, 2, device=DEV); ob = torch.empty(M, H, dtype=BF16, device=DEV)
    L.ln_fwd(y=t3, gamma=g2, beta=b2, eps=eps, out_stats=st, out_bf16=ob, N=M, H=H)
    close(st[:, 0], st_ref[:, 0], 1e-4, 1e-4, "mean")
    close(st[:, 1], st_ref[:, 1], 1e-3, 1e-4, "rstd")
    close(ob, ob_ref, 2e-2, 2e-2, "bf16 operand")
    # argument contract: the generic GEMM entry refuses the internal epilogue kind
    with pytest.raises(RuntimeError):
        L.gemm(z, Wu, bias=bu, aux=x, aux_kind=6, out_f32=t)
