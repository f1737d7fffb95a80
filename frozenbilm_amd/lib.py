"""ctypes binding of libfbl.so (include/fbl.h) + thin tensor-level wrappers.

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``) and the current HIP stream.  There is
NO fallback: if the shared library is missing or a kernel returns an error, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# FBL_LIB: measurement tools point this at libfbl_dbg.so (FBL_DEBUG_BUILD=1 build); the product never sets it
LIB_PATH = os.environ.get("FBL_LIB") or os.path.join(HERE, "libfbl.so")

ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_GRAD = 0, 1, 2, 3
AUX_NONE, AUX_ADD_F32, AUX_ADD_BF16, AUX_MUL_DGELU_BF16, AUX_MUL_POS_BF16, AUX_MUL_BF16 = 0, 1, 2, 3, 4, 5
ABI_VERSION = 8  # fbl_abi_version() of the library this binding was written against (argument lists change with it)

_vp, _i, _l, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); must mirror include/fbl.h (tests/test_abi.py checks both directions)
SIGNATURES = {
    "fbl_abi_version": (_i, []),
    "fbl_gemm_bf16_nt": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _f, _i, _i, _vp, _l, _vp, _vp, _vp, _l, _i, _l, _l,
                              _l, _l, _l, _i, _vp, _l, _l, _vp, _i, _vp, _vp, _vp]),
    "fbl_adapter_down_fwd": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _f, _u64, _vp, _vp, _l, _vp]),
    "fbl_dense_adapter_down_fwd": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _i, _vp, _vp, _vp, _l, _f, _u64, _vp, _vp, _l, _vp, _vp]),
    "fbl_adapter_up_resid_fwd": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _l, _f, _u64, _vp, _vp, _l, _vp, _vp, _vp, _vp,
                                      _vp, _l, _vp]),
    "fbl_gemm_bf16_tn_acc": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _l, _i, _vp, _l, _vp]),
    "fbl_adapter_bwd_dw": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fbl_embed_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fbl_gemm_plan": (_i, [_i, _i, _i, _i, _i]),
    "fbl_ln_fwd": (_i, [_vp, _l, _f, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i,
                        _vp]),
    "fbl_ln_materialize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "fbl_ln_bwd_ws_floats": (_l, [_i]),
    "fbl_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp]),
    "fbl_im2col3": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "fbl_col2im3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "fbl_dropout_gelu_fwd": (_i, [_vp, _f, _u64, _vp, _vp, _l, _vp]),
    "fbl_dropout_gelu_bwd": (_i, [_vp, _vp, _f, _u64, _vp, _vp, _vp, _l, _vp]),
    "fbl_video_stage_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fbl_mask_tokens": (_i, [_vp, _vp, _l, _vp, _i, _f, _l, _l, _u64, _vp]),
    "fbl_transpose_to_bf16": (_i, [_vp, _i, _l, _i, _i, _vp, _l, _vp]),
    "fbl_transpose_batched_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "fbl_colsum_ws_floats": (_l, [_i]),
    "fbl_colsum": (_i, [_vp, _i, _l, _i, _i, _vp, _vp, _vp]),
    "fbl_head_transpose": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _l, _l, _l, _vp]),
    "fbl_disent_attn_fwd": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _f, _f, _u64, _vp, _vp, _l,
                                 _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fbl_disent_attn_probs": (_i, [_vp, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "fbl_attn_rowdot": (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _vp]),
    "fbl_attn_bwd_prep": (_i, [_vp, _vp, _l, _vp, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "fbl_disent_attn_bwd_dq": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp, _vp]),
    "fbl_disent_attn_bwd_dspk": (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _vp, _f, _f, _u64, _vp, _vp, _l, _vp, _l,
                                      _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fbl_disent_attn_bwd_ds": (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _f, _f, _u64, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "fbl_disent_attn_bwd_dsp": (_i, [_vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _f, _f, _u64, _vp, _vp, _l, _vp, _vp,
                                     _i, _i, _i, _i, _vp, _vp]),
    "fbl_disent_attn_bwd_shear": (_i, [_i, _vp, _vp, _l, _l, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                       _vp, _vp, _vp]),
    "fbl_attn_pos_grad": (_i, [_i, _vp, _vp, _l, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fbl_gt_tilemask": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "fbl_ce_fwd": (_i, [_vp, _l, _vp, _i, _i, _vp, _vp, _vp]),
    "fbl_ce_bwd_rows": (_i, [_vp, _l, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "fbl_gather_rows_bf16": (_i, [_vp, _l, _vp, _i, _i, _vp, _vp]),
    "fbl_scatter_rows_f32": (_i, [_vp, _vp, _i, _i, _vp, _l, _vp]),
    "fbl_sumsq_ws_floats": (_l, []),
    "fbl_sumsq": (_i, [_vp, _l, _vp, _vp, _vp]),
    "fbl_adam_flat": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp]),
    "fbl_cast_f32_to_bf16": (_i, [_vp, _vp, _l, _vp]),
    "fbl_dropout_f32": (_i, [_vp, _f, _u64, _vp, _vp, _vp, _l, _vp]),
    "fbl_dropout_bf16": (_i, [_vp, _f, _u64, _vp, _l, _vp]),
    "fbl_dropout_sum_f32": (_i, [_vp, _l, _l, _i, _vp, _f, _vp, _vp, _vp]),
    "fbl_zero": (_i, [_vp, _l, _vp]),
    "fbl_heads_to_rows_bf16": (_i, [_vp, _vp, _i, _i, _i, _l, _vp]),
}

_LIB = None


def load(path: Optional[str] = None):
    """Load libfbl.so and declare the ABI.  Raises if the library or any declared symbol is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the FrozenBiLM MI355X path needs its HIP library (python -m frozenbilm_amd.build); "
            "there is no CPU/eager fallback")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.fbl_abi_version()
    if ver != ABI_VERSION:  # same symbol names, different argument lists: calling through would shift arguments silently
        raise RuntimeError(f"{p} has ABI version {ver}, this binding needs {ABI_VERSION}: rebuild it "
                           "(python -m frozenbilm_amd.build --force)")
    if path is None:
        _LIB = lib
    return lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# The caller-provided aux stream of the two GEMM entry points that can use one (include/fbl.h).  The library owns no
# stream: the engine creates one per device and registers it here.  The bindings pass it for launches on any stream of
# that device (also a capturing one: the fork / join lands in the capture) EXCEPT the aux stream itself and the streams
# listed in _NO_AUX -- the engine's side stream: its GEMMs must not fork into the in-order aux stream the main stream's
# GEMMs use (that would make side work wait for main-stream remainder tiles and, under capture, pull it into the graph).
_AUX = {}
_NO_AUX = set()


def exclude_from_aux(stream: "torch.cuda.Stream"):
    """launches on `stream` never get the aux stream (see above)"""
    _NO_AUX.add(stream.cuda_stream)



def set_aux_stream(stream: Optional["torch.cuda.Stream"], device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if stream is None:
        _AUX.pop(idx, None)
    else:
        _AUX[idx] = stream


def _aux_stream():
    st = _AUX.get(torch.cuda.current_device())
    if st is None:
        return None
    h = st.cuda_stream
    cur = torch.cuda.current_stream().cuda_stream
    return None if (h == cur or cur in _NO_AUX) else h


# The device-resident seed word (include/fbl.h "Dropout seeds"): while a word is set -- by the engine, for the launches of one
# forward / backward pass -- every seeded launch made through this binding passes it as `seed_dev`; otherwise NULL.
_SEED_WORD = None


class seed_word:
    """context manager: seeded launches inside it add the 64-bit device word `t` (int64 tensor, 1 element) to their seed"""

    def __init__(self, t: Optional[torch.Tensor]):
        self.t, self.prev = t, None

    def __enter__(self):
        global _SEED_WORD
        if self.t is not None:
            assert self.t.dtype == torch.int64 and self.t.is_cuda and self.t.numel() >= 1
        self.prev, _SEED_WORD = _SEED_WORD, self.t
        return self

    def __exit__(self, *exc):
        global _SEED_WORD
        _SEED_WORD = self.prev
        return False


def _seed_dev():
    return None if _SEED_WORD is None else _SEED_WORD.data_ptr()


def _chk(code: int, name: str):
    if code != 0:
        raise RuntimeError(f"{name} failed with code {code}")


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live in HBM (cuda device), got {t.device}")


def _rows2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D row-major view (stride(1)==1), got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_plan(M, N, K, batch=1, splitk=1):
    """8 if fbl_gemm_bf16_nt runs this plain problem on the 8-phase kernel (gemm8_kernel), 2 for the 2-stage kernel"""
    return load().fbl_gemm_plan(int(M), int(N), int(K), int(batch), int(splitk))


def gemm(A, B, *, bias=None, rowscale=None, alpha=1.0, act=ACT_NONE, aux=None, aux_kind=AUX_NONE, out_f32=None,
         out_bf16=None, out_pre=None, splitk=1, M=None, N=None, ws=None, K=None, a_kblock=0, kskip_len=None, kskip_steps=0,
         kskip_tilemask=None):
    """out[M,N] = epi(alpha * A[M,K] @ B[N,K]^T).  A/B: bf16 2-D views (or 3-D for strided batch)."""
    _req(A, torch.bfloat16, "A")
    _req(B, torch.bfloat16, "B")
    batch = 1
    sA = sB = sC = sX = 0
    if A.dim() == 3:
        batch = A.shape[0]
        sA, sB = A.stride(0), (B.stride(0) if B.dim() == 3 else 0)
        A2, B2 = A[0], (B[0] if B.dim() == 3 else B)
    else:
        A2, B2 = A, B
    lda, ldb = _rows2d(A2, "A"), _rows2d(B2, "B")
    Mv, Kv = A2.shape
    Nv = B2.shape[0]
    if K is None:
        K = Kv
        assert B2.shape[1] == K, (A2.shape, B2.shape)
    M = Mv if M is None else M
    N = Nv if N is None else N
    ldc = None
    for o, dt in ((out_f32, torch.float32), (out_bf16, torch.bfloat16), (out_pre, torch.bfloat16)):
        if o is None:
            continue
        _req(o, dt, "out")
        o2 = o[0] if o.dim() == 3 else o
        l = _rows2d(o2, "out")
        assert o2.shape[0] >= M and o2.shape[1] >= N, (o2.shape, M, N)
        if ldc is None:
            ldc = l
            sC = o.stride(0) if o.dim() == 3 else 0
        assert ldc == l, "all outputs of one GEMM must share the row stride"
    ld_aux = 0
    if aux is not None:
        a2 = aux[0] if aux.dim() == 3 else aux
        ld_aux = _rows2d(a2, "aux")
        sX = aux.stride(0) if aux.dim() == 3 else 0
        _req(aux, torch.float32 if aux_kind == AUX_ADD_F32 else torch.bfloat16, "aux")
    sBias = 0
    if bias is not None:
        _req(bias, torch.float32, "bias")
        if bias.dim() == 2:
            sBias = bias.stride(0)
    if rowscale is not None:
        _req(rowscale, torch.float32, "rowscale")
    code = load().fbl_gemm_bf16_nt(_p(A), lda, _p(B), ldb, M, N, K, _p(bias), _p(rowscale), float(alpha), act, aux_kind,
                                   _p(aux), ld_aux, _p(out_f32), _p(out_bf16), _p(out_pre), ldc or 0, batch, sA, sB, sC,
                                   sX, sBias, splitk, _p(ws), (ws.numel() if ws is not None else 0), int(a_kblock),
                                   _p(kskip_len), int(kskip_steps), _p(kskip_tilemask), _stream(), _aux_stream())
    _chk(code, "fbl_gemm_bf16_nt")


def adapter_down_fwd(x, wd, bias, z, *, A=None, p_drop=0.0, seed=0):
    """z[M, A] = dropout(relu(x @ wd[:A]^T + bias)): one launch (ReLU and dropout live in the GEMM epilogue)."""
    _req(x, torch.bfloat16, "x"); _req(wd, torch.bfloat16, "wd"); _req(z, torch.bfloat16, "z")
    _req(bias, torch.float32, "bias")
    ldx, ldw, ldz = _rows2d(x, "x"), _rows2d(wd, "wd"), _rows2d(z, "z")
    M, K = x.shape
    A = wd.shape[0] if A is None else A
    assert wd.shape[1] == K and z.shape[0] >= M and z.shape[1] >= A
    _chk(load().fbl_adapter_down_fwd(_p(x), ldx, _p(wd), ldw, M, A, K, _p(bias), float(p_drop), int(seed), _seed_dev(), _p(z), ldz,
                                     _stream()), "fbl_adapter_down_fwd")


def dense_adapter_down_fwd(x, wm, bias_m, N1, z, *, y_f32=None, y_bf16=None, p_drop=0.0, seed=0):
    """[y | z] from one GEMM: y = x @ wm[:N1]^T + bias_m[:N1] (fp32 / bf16), z = dropout(relu(x @ wm[N1:]^T + bias_m[N1:]))."""
    _req(x, torch.bfloat16, "x"); _req(wm, torch.bfloat16, "wm"); _req(z, torch.bfloat16, "z"); _req(bias_m, torch.float32, "bias")
    ldx, ldw, ldz = _rows2d(x, "x"), _rows2d(wm, "wm"), _rows2d(z, "z")
    M, K = x.shape
    A = wm.shape[0] - N1
    assert wm.shape[1] == K and A > 0 and z.shape[0] >= M and z.shape[1] >= A and bias_m.numel() >= N1 + A
    ldy = None
    for o, dt in ((y_f32, torch.float32), (y_bf16, torch.bfloat16)):
        if o is not None:
            _req(o, dt, "y")
            l = _rows2d(o, "y")
            assert ldy is None or ldy == l
            ldy = l
    _chk(load().fbl_dense_adapter_down_fwd(_p(x), ldx, _p(wm), ldw, M, N1, A, K, _p(bias_m), _p(y_f32), _p(y_bf16), ldy or 0,
                                           float(p_drop), int(seed), _seed_dev(), _p(z), ldz, _stream(), _aux_stream()),
         "fbl_dense_adapter_down_fwd")


def adapter_up_resid_fwd(z, wu, bias_u, x, out_t, *, A=None, p_drop=0.0, seed=0, r_plain=None, r_norm=None):
    """out_t[M,H] = dropout(x + z[:, :A] @ wu[:, :A]^T + bias_u) + residual (fp32): the adapter's up-projection with the
    residual / dropout of the enclosing block in its epilogue.  residual: r_plain (fp32 [M,H]) or r_norm = (t, stats, gamma,
    beta, rowmask|None), the LayerNorm-normalised form of fbl_ln_fwd.  Dropout keys as fbl_ln_fwd (seed, m*H + n)."""
    _req(z, torch.bfloat16, "z"); _req(wu, torch.bfloat16, "wu"); _req(x, torch.bfloat16, "x"); _req(out_t, torch.float32, "out_t")
    ldz, ldw, ldx, ldt = _rows2d(z, "z"), _rows2d(wu, "wu"), _rows2d(x, "x"), _rows2d(out_t, "out_t")
    M, H = x.shape
    A = wu.shape[1] if A is None else A
    assert wu.shape[0] == H and z.shape[0] >= M and z.shape[1] >= A and out_t.shape[0] >= M and out_t.shape[1] >= H
    if bias_u is not None:
        _req(bias_u, torch.float32, "bias_u")
    if (r_plain is None) == (r_norm is None):
        raise ValueError("exactly one of r_plain / r_norm")
    rs = rg = rb = rm = None
    if r_norm is not None:
        rt, rs, rg, rb, rm = r_norm
        for t in (rt, rs, rg, rb):
            _req(t, torch.float32, "r_norm")
        assert rs.is_contiguous() and rs.numel() >= 2 * M
        if rm is not None:
            _req(rm, torch.int32, "rowmask")
    else:
        rt = r_plain
        _req(rt, torch.float32, "r_plain")
    ld_r = _rows2d(rt, "residual")
    assert rt.shape[0] >= M and rt.shape[1] >= H
    _chk(load().fbl_adapter_up_resid_fwd(_p(z), ldz, _p(wu), ldw, M, H, int(A), _p(bias_u), _p(x), ldx, float(p_drop),
                                         int(seed), _seed_dev(), _p(rt), ld_r, _p(rs), _p(rg), _p(rb), _p(rm), _p(out_t), ldt,
                                         _stream()), "fbl_adapter_up_resid_fwd")


def gemm_tn_acc(A, B, out_f32, ws, *, M=None, N=None, K=None, splitk=8):
    """out[M,N] += A[:K,:M]^T @ B[:K,:N]  (A [K,M'], B [K,N'] row-major bf16; contraction over rows)."""
    _req(A, torch.bfloat16, "A"); _req(B, torch.bfloat16, "B"); _req(out_f32, torch.float32, "out")
    lda, ldb, ldc = _rows2d(A, "A"), _rows2d(B, "B"), _rows2d(out_f32, "out")
    K = min(A.shape[0], B.shape[0]) if K is None else K
    M = A.shape[1] if M is None else M
    N = B.shape[1] if N is None else N
    assert out_f32.shape[0] >= M and out_f32.shape[1] >= N
    _chk(load().fbl_gemm_bf16_tn_acc(_p(A), lda, _p(B), ldb, M, N, K, _p(out_f32), ldc, splitk, _p(ws), ws.numel(),
                                     _stream()), "fbl_gemm_bf16_tn_acc")


ADW_MAX_ADAPTERS, ADW_MAX_SEGMENTS = 16, 24  # include/fbl.h


def adapter_bwd_dw(groups, *, A):
    """One launch for a group of same-shaped adapters.  groups: list of (segments, dWu, dWd, dbd), segments a list of
    (dy [N,H], z [N,Ap], dz [N,Ap], x [N,H]) bf16 -- one per execution of that adapter in the forward pass;
    dWu[H,A] += sum dy^T z, dWd[A,H] += sum dz^T x, dbd[A] += sum colsum(dz) (any of the three may be None)."""
    segs = [sg for g in groups for sg in g[0]]
    assert 0 < len(groups) <= ADW_MAX_ADAPTERS and 0 < len(segs) <= ADW_MAX_SEGMENTS
    N, H = segs[0][0].shape
    Ap = segs[0][1].shape[1]
    strides = []
    for g in groups:
        lds = None
        for dy, z, dz, x in g[0]:
            for t, n in ((dy, "dy"), (z, "z"), (dz, "dz"), (x, "x")):
                _req(t, torch.bfloat16, n)
            assert dy.shape == (N, H) and x.shape == (N, H) and z.shape == (N, Ap) and dz.shape == (N, Ap) and Ap >= A
            l = (_rows2d(dy, "dy"), _rows2d(z, "z"), _rows2d(dz, "dz"), _rows2d(x, "x"))
            assert lds is None or l == lds, "the segments of an adapter share their row strides"
            lds = l
        strides.append(lds)
    for _, dWu, dWd, dbd in groups:
        for o, shp in ((dWu, (H, A)), (dWd, (A, H)), (dbd, (A,))):
            if o is not None:
                _req(o, torch.float32, "out")
                assert tuple(o.shape) == shp and o.is_contiguous(), (tuple(o.shape), shp)
    first = [0]
    for g in groups:
        first.append(first[-1] + len(g[0]))
    seg_first = (C.c_int32 * len(first))(*first)
    tab = lambda ts: (C.c_void_p * len(ts))(*[_p(t) for t in ts])
    ldt = lambda k: (C.c_int64 * len(strides))(*[st[k] for st in strides])
    _chk(load().fbl_adapter_bwd_dw(len(groups), seg_first, tab([s[0] for s in segs]), tab([s[1] for s in segs]),
                                   tab([s[2] for s in segs]), tab([s[3] for s in segs]), ldt(0), ldt(1), ldt(2), ldt(3),
                                   N, H, int(A), Ap, tab([g[1] for g in groups]), tab([g[2] for g in groups]),
                                   tab([g[3] for g in groups]), _stream()), "fbl_adapter_bwd_dw")


# ------------------------------------------------------------------------------------------------ row ops
def embed_gather(ids, E, vproj, T, out_t):
    B, L = ids.shape
    H = E.shape[1]
    _req(ids, torch.int64, "ids"); _req(E, torch.float32, "E"); _req(out_t, torch.float32, "out_t")
    assert ids.is_contiguous() and E.is_contiguous() and out_t.is_contiguous()
    if vproj is not None:
        _req(vproj, torch.float32, "vproj")
        assert vproj.is_contiguous()
    _chk(load().fbl_embed_gather(_p(ids), _p(E), _p(vproj), B, T, L, H, _p(out_t), _stream()), "fbl_embed_gather")


def ln_fwd(*, y=None, p_drop=0.0, seed=0, r_plain=None, r_norm=None, gamma, beta, eps, rowmask=None, out_t=None,
           out_stats=None, out_bf16=None, out_f32=None, N, H):
    """r_norm = (t, stats, gamma, beta, rowmask|None): residual given in LayerNorm-normalised form."""
    ldy = 0
    if y is not None:
        _req(y, torch.float32, "y")
        ldy = _rows2d(y, "y")
    rt = rs = rg = rb = rm = None
    if r_norm is not None:
        rt, rs, rg, rb, rm = r_norm
    for t in (r_plain, rt, out_t, out_bf16, out_f32):
        if t is not None:
            assert t.is_contiguous()
    _chk(load().fbl_ln_fwd(_p(y), ldy, float(p_drop), int(seed), _seed_dev(), _p(r_plain), _p(rt), _p(rs), _p(rg), _p(rb), _p(rm),
                           _p(gamma), _p(beta), float(eps), _p(rowmask), _p(out_t), _p(out_stats), _p(out_bf16),
                           _p(out_f32), N, H, _stream()), "fbl_ln_fwd")


def ln_materialize(t, stats, gamma, beta, rowmask=None, add_bcast=None, S=1, out_f32=None, out_bf16=None):
    N, H = t.shape
    _chk(load().fbl_ln_materialize(_p(t), _p(stats), _p(gamma), _p(beta), _p(rowmask), _p(add_bcast), S, _p(out_f32),
                                   _p(out_bf16), N, H, _stream()), "fbl_ln_materialize")


def ln_bwd_ws(H, device):
    return torch.empty(load().fbl_ln_bwd_ws_floats(H), dtype=torch.float32, device=device)


def ln_bwd(dout, t, stats, gamma, *, rowmask=None, p_drop=0.0, seed=0, out_dt=None, out_dy_bf16=None, out_dy_f32=None,
           dgamma=None, dbeta=None, dysum=None, ws=None):
    N, H = t.shape
    assert dout.is_contiguous() and t.is_contiguous()
    ld_dyb = _rows2d(out_dy_bf16, "out_dy_bf16") if out_dy_bf16 is not None else 0
    _chk(load().fbl_ln_bwd(_p(dout), _p(rowmask), _p(t), _p(stats), _p(gamma), float(p_drop), int(seed), _seed_dev(), _p(out_dt),
                           _p(out_dy_bf16), _p(out_dy_f32), _p(dgamma), _p(dbeta), _p(dysum), _p(ws), N, H, ld_dyb,
                           _stream()), "fbl_ln_bwd")


def im2col3(x_bf16, out_bf16, B, S, H):
    _chk(load().fbl_im2col3(_p(x_bf16), _p(out_bf16), B, S, H, _stream()), "fbl_im2col3")


def col2im3(dcol, dx, B, S, H, accumulate):
    _chk(load().fbl_col2im3(_p(dcol), _p(dx), B, S, H, int(accumulate), _stream()), "fbl_col2im3")


def dropout_gelu_fwd(c, p_drop, seed, out_f32):
    _chk(load().fbl_dropout_gelu_fwd(_p(c), float(p_drop), int(seed), _seed_dev(), _p(out_f32), c.numel(), _stream()),
         "fbl_dropout_gelu_fwd")


def dropout_gelu_bwd(dy, c, p_drop, seed, out_bf16=None, out_f32=None):
    _chk(load().fbl_dropout_gelu_bwd(_p(dy), _p(c), float(p_drop), int(seed), _seed_dev(), _p(out_bf16), _p(out_f32), c.numel(),
                                     _stream()), "fbl_dropout_gelu_bwd")


def video_stage_f16(feats, row_off, n_rows, T):
    """packed fp16 clips -> (video fp32 [B,T,F], video_len int64 [B], video_mask int64 [B,T])"""
    assert feats.dtype == torch.float16 and feats.is_contiguous() and row_off.dtype == torch.int64 and n_rows.dtype == torch.int32
    B, F_ = n_rows.numel(), feats.shape[1]
    video = torch.empty(B, T, F_, dtype=torch.float32, device=feats.device)
    vlen = torch.empty(B, dtype=torch.int64, device=feats.device)
    vmask = torch.empty(B, T, dtype=torch.int64, device=feats.device)
    _chk(load().fbl_video_stage_f16(_p(feats), _p(row_off), _p(n_rows), B, T, F_, _p(video), _p(vlen), _p(vmask), _stream()),
         "fbl_video_stage_f16")
    return video, vlen, vmask


def mask_tokens(ids, labels, special_ids, p, mask_token_id, vocab_size, seed):
    assert ids.dtype == torch.int64 and labels.dtype == torch.int64 and ids.is_contiguous() and labels.is_contiguous()
    assert special_ids.dtype == torch.int64
    _chk(load().fbl_mask_tokens(_p(ids), _p(labels), ids.numel(), _p(special_ids), special_ids.numel(), float(p),
                                int(mask_token_id), int(vocab_size), int(seed), _stream()), "fbl_mask_tokens")


def transpose_batched_bf16(src, src_off, dst, dst_off, rows, cols):
    """dst[dst_off[i]:][cols, rows] = src[src_off[i]:][rows, cols]^T for every i; offsets: int64 device tensors (elements)."""
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16
    assert src_off.dtype == torch.int64 and dst_off.dtype == torch.int64 and src_off.numel() == dst_off.numel()
    _chk(load().fbl_transpose_batched_bf16(_p(src), _p(src_off), _p(dst), _p(dst_off), src_off.numel(), rows, cols,
                                           _stream()), "fbl_transpose_batched_bf16")


def transpose_to_bf16(x, out, rows=None, cols=None):
    """out[c, r] = x[r, c]; out is [cols, rows_pad] bf16 (zero padded)."""
    ld = _rows2d(x, "x")
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    assert out.is_contiguous() and out.shape[0] >= cols
    _req(out, torch.bfloat16, "out")
    is_bf = 1 if x.dtype == torch.bfloat16 else 0
    if not is_bf:
        _req(x, torch.float32, "x")
    _chk(load().fbl_transpose_to_bf16(_p(x), is_bf, ld, rows, cols, _p(out), out.shape[1], _stream()),
         "fbl_transpose_to_bf16")


def colsum_ws(cols, device):
    return torch.empty(load().fbl_colsum_ws_floats(cols), dtype=torch.float32, device=device)


def colsum(x, out, ws, rows=None, cols=None):
    ld = _rows2d(x, "x")
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    is_bf = 1 if x.dtype == torch.bfloat16 else 0
    _chk(load().fbl_colsum(_p(x), is_bf, ld, rows, cols, _p(out), _p(ws), _stream()), "fbl_colsum")


def head_strides(B, Sp, nh, head_major):
    """(sh, sb, sd) of a per-head transposed tensor: [B,nh,64,Sp] (head_major=False) or [nh,64,B,Sp] (True)."""
    if head_major:
        return 64 * B * Sp, Sp, B * Sp
    return 64 * Sp, nh * 64 * Sp, Sp


def head_transpose(v, vt, B, S, Sp, nh, head_major=False):
    ldv = _rows2d(v, "v")
    _req(v, torch.bfloat16, "v"); _req(vt, torch.bfloat16, "vt")
    assert vt.is_contiguous() and vt.numel() == B * nh * 64 * Sp
    sh, sb, sd = head_strides(B, Sp, nh, head_major)
    _chk(load().fbl_head_transpose(_p(v), ldv, _p(vt), B, S, Sp, nh, sh, sb, sd, _stream()), "fbl_head_transpose")


def _row0(row0, B, klen):
    """packed-row layout argument of the attention entry points: int32 [B+1] row offsets (needs klen) or None"""
    if row0 is None:
        return None
    _req(row0, torch.int32, "row0")
    assert row0.is_contiguous() and row0.numel() == B + 1 and klen is not None
    return _p(row0)


def disent_attn_fwd(q, k, v, pk, pq, relidx, mask, scale, ctx, lse, B, S, Sp, nh, span2, p_drop=0.0, seed=0, klen=None,
                    border=None, lin=0, row0=None, psave=None, msave=None):
    """psave bf16 [B,nh,Sp,Sp] + msave fp32 [B,nh,Sp/64,S] (training): the forward leaves its un-normalised probabilities
    for fbl_disent_attn_bwd_dsp (include/fbl.h)"""
    if psave is not None:
        _req(psave, torch.bfloat16, "psave"); _req(msave, torch.float32, "msave")
        assert psave.is_contiguous() and msave.is_contiguous()
        assert psave.numel() == B * nh * Sp * Sp and msave.numel() == B * nh * (Sp // 64) * S
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (pk, "pk"), (pq, "pq"), (ctx, "ctx")):
        _req(t, torch.bfloat16, n)
    _req(relidx, torch.int16, "relidx"); _req(mask, torch.int32, "mask")
    assert relidx.numel() == 2 * S - 1 and mask.is_contiguous()
    ldq, ldk, ldv, ldp, ldo = _rows2d(q, "q"), _rows2d(k, "k"), _rows2d(v, "v"), _rows2d(pk, "pk"), _rows2d(ctx, "ctx")
    assert _rows2d(pq, "pq") == ldp
    _chk(load().fbl_disent_attn_fwd(_p(q), ldq, _p(k), ldk, _p(v), ldv, _p(pk), _p(pq), ldp, _p(relidx),
                                    _p(mask), _p(klen), _p(border), float(scale), float(p_drop), int(seed), _seed_dev(), _p(ctx), ldo,
                                    _p(lse), B, S, Sp, nh, span2, int(lin), _row0(row0, B, klen), _p(psave), _p(msave), _stream()),
         "fbl_disent_attn_fwd")


def disent_attn_probs(q, k, pk, pq, relidx, mask, lse, scale, probs, B, S, nh):
    """probs[B,nh,S,S] (fp32) = the attention probabilities behind fbl_disent_attn_fwd's ctx (from its lse); eval mode"""
    for t, n in ((q, "q"), (k, "k"), (pk, "pk"), (pq, "pq")):
        _req(t, torch.bfloat16, n)
    _req(relidx, torch.int16, "relidx"); _req(mask, torch.int32, "mask"); _req(lse, torch.float32, "lse"); _req(probs, torch.float32, "probs")
    ldq, ldp = _rows2d(q, "q"), _rows2d(pk, "pk")
    assert _rows2d(k, "k") == ldq and _rows2d(pq, "pq") == ldp and probs.is_contiguous() and probs.numel() == B * nh * S * S
    _chk(load().fbl_disent_attn_probs(_p(q), _p(k), ldq, _p(pk), _p(pq), ldp, _p(relidx), _p(mask), _p(lse), float(scale),
                                      _p(probs), B, S, nh, _stream()), "fbl_disent_attn_probs")


def attn_bwd_prep(q, k, pq, pk, dO, O, QT, KT, PQT, PKT, Dv, B, S, Sp, nh, span2, row0=None, relidx=None, PQX=None, PKX=None):
    """K^T, Q^T (head-major), PK^T, PQ^T, D = rowdot(dO, O) and the index-expanded tables PQX / PKX [nh,2*Sp,64] in one launch
    (see fbl.h); every output but Dv may be None"""
    ldq, ldp, ldo = _rows2d(q, "q"), _rows2d(pq, "pq"), _rows2d(dO, "dO")
    assert _rows2d(k, "k") == ldq and _rows2d(pk, "pk") == ldp and _rows2d(O, "O") == ldo
    for t in (QT, KT, PQT, PKT, PQX, PKX):
        if t is not None:
            _req(t, torch.bfloat16, "transposed output")
            assert t.is_contiguous()
    for t in (PQX, PKX):
        if t is not None:
            _req(relidx, torch.int16, "relidx")
            assert t.numel() == nh * 64 * 2 * Sp and relidx.numel() == 2 * S - 1
    _chk(load().fbl_attn_bwd_prep(_p(q), _p(k), ldq, _p(pq), _p(pk), ldp, _p(dO), _p(O), ldo, _p(QT), _p(KT), _p(PQT), _p(PKT),
                                  _p(Dv), _p(relidx), _p(PQX), _p(PKX), B, S, Sp, nh, span2, _row0(row0, B, True), _stream()),
         "fbl_attn_bwd_prep")


def attn_rowdot(dO, O, out, B, S, nh):
    ld = _rows2d(dO, "dO")
    assert _rows2d(O, "O") == ld
    _chk(load().fbl_attn_rowdot(_p(dO), _p(O), ld, _p(out), B, S, nh, _stream()), "fbl_attn_rowdot")


def disent_attn_bwd_ds(q, k, v, dO, pk, pq, relidx, mask, lse, Dv, scale, dV, dS, dST, B, S, Sp, nh, span2,
                       p_drop=0.0, seed=0, klen=None, border=None, lin=0, row0=None):
    ldq = _rows2d(q, "q")
    assert _rows2d(k, "k") == ldq and _rows2d(v, "v") == ldq
    ldo, ldp, lddv = _rows2d(dO, "dO"), _rows2d(pk, "pk"), _rows2d(dV, "dV")
    assert _rows2d(pq, "pq") == ldp
    _chk(load().fbl_disent_attn_bwd_ds(_p(q), _p(k), _p(v), ldq, _p(dO), ldo, _p(pk), _p(pq), ldp,
                                       _p(relidx), _p(mask), _p(klen), _p(border), _p(lse), _p(Dv), float(scale), float(p_drop), int(seed), _seed_dev(),
                                       _p(dV), lddv, _p(dS), _p(dST), B, S, Sp, nh, span2, int(lin), _row0(row0, B, klen), _stream()),
         "fbl_disent_attn_bwd_ds")


def disent_attn_bwd_dsp(psave, msave, v, dO, lse, Dv, scale, dV, dS, dST, B, S, Sp, nh, p_drop=0.0, seed=0, klen=None,
                        border=None, row0=None):
    """kernel A of the attention backward from the probabilities the training forward saved (no recomputation of the scores)"""
    _req(psave, torch.bfloat16, "psave"); _req(msave, torch.float32, "msave")
    assert psave.is_contiguous() and msave.is_contiguous()
    assert psave.numel() == B * nh * Sp * Sp and msave.numel() == B * nh * (Sp // 64) * S
    ldv, ldo, lddv = _rows2d(v, "v"), _rows2d(dO, "dO"), _rows2d(dV, "dV")
    _chk(load().fbl_disent_attn_bwd_dsp(_p(psave), _p(msave), _p(v), ldv, _p(dO), ldo, _p(klen), _p(border), _p(lse), _p(Dv),
                                        float(scale), float(p_drop), int(seed), _seed_dev(), _p(dV), lddv, _p(dS), _p(dST),
                                        B, S, Sp, nh, _row0(row0, B, klen), _stream()), "fbl_disent_attn_bwd_dsp")


def disent_attn_bwd_dspk(psave, msave, q, v, dO, pqx, lse, Dv, scale, dK, dV, dS, dST, B, S, Sp, nh, p_drop=0.0, seed=0, klen=None,
                         border=None, row0=None):
    """kernel A from the saved probabilities with dK = dS^T.Q + G2.PQ formed in place (fbl_disent_attn_bwd_dspk, include/fbl.h)"""
    _req(psave, torch.bfloat16, "psave"); _req(msave, torch.float32, "msave"); _req(pqx, torch.bfloat16, "pqx")
    assert psave.is_contiguous() and msave.is_contiguous() and pqx.is_contiguous()
    assert psave.numel() == B * nh * Sp * Sp and msave.numel() == B * nh * (Sp // 64) * S and pqx.numel() == nh * 64 * 2 * Sp
    ldq, ldv, ldo, lddk, lddv = _rows2d(q, "q"), _rows2d(v, "v"), _rows2d(dO, "dO"), _rows2d(dK, "dK"), _rows2d(dV, "dV")
    _chk(load().fbl_disent_attn_bwd_dspk(_p(psave), _p(msave), _p(q), ldq, _p(v), ldv, _p(dO), ldo, _p(pqx), _p(klen), _p(border),
                                         _p(lse), _p(Dv), float(scale), float(p_drop), int(seed), _seed_dev(), _p(dK), lddk,
                                         _p(dV), lddv, _p(dS), _p(dST), B, S, Sp, nh, _row0(row0, B, klen), _stream()),
         "fbl_disent_attn_bwd_dspk")


def disent_attn_bwd_dq(dS, k, pkx, dQ, B, S, Sp, nh, klen=None, border=None, row0=None):
    """dQ = dS.K + G1.PK in Toeplitz form (fbl_disent_attn_bwd_dq, include/fbl.h)"""
    _req(dS, torch.bfloat16, "dS"); _req(pkx, torch.bfloat16, "pkx")
    assert dS.is_contiguous() and pkx.is_contiguous() and dS.numel() == B * nh * Sp * Sp and pkx.numel() == nh * 2 * Sp * 64
    ldk, lddq = _rows2d(k, "k"), _rows2d(dQ, "dQ")
    _chk(load().fbl_disent_attn_bwd_dq(_p(dS), _p(k), ldk, _p(pkx), _p(klen), _p(border), _p(dQ), lddq, B, S, Sp, nh,
                                       _row0(row0, B, klen), _stream()), "fbl_disent_attn_bwd_dq")


def attn_pos_grad(neg, Xs, Ys, dlo, dcnt, dcnt_max, out, B, S, Sp, nh, rcnt, klen=None, row0=None):
    """out[e, h, r, :] fp32 = position-table gradient (neg=0: dPK from dS and q; neg=1: dPQ from dS^T and k) of the layer
    executions whose tensors are listed in Xs / Ys (include/fbl.h fbl_attn_pos_grad)"""
    E = len(Xs)
    assert E == len(Ys) and E > 0
    ldy = _rows2d(Ys[0], "Y")
    for x, y in zip(Xs, Ys):
        _req(x, torch.bfloat16, "X"); _req(y, torch.bfloat16, "Y")
        assert x.is_contiguous() and x.numel() == B * nh * Sp * Sp and _rows2d(y, "Y") == ldy
    _req(dlo, torch.int16, "dlo"); _req(dcnt, torch.int16, "dcnt"); _req(out, torch.float32, "out")
    assert dlo.numel() == rcnt and dcnt.numel() == rcnt and out.is_contiguous() and out.numel() == E * nh * rcnt * 64
    xp = (C.c_void_p * E)(*[x.data_ptr() for x in Xs])
    yp = (C.c_void_p * E)(*[y.data_ptr() for y in Ys])
    _chk(load().fbl_attn_pos_grad(int(neg), xp, yp, ldy, _p(dlo), _p(dcnt), int(dcnt_max), _p(klen), _row0(row0, B, klen), _p(out),
                                  E, B, S, Sp, nh, int(rcnt), _stream()), "fbl_attn_pos_grad")


def gt_tilemask(relidx, klen, B, S, Sp, span2, neg, rmin, rcnt):
    """uint32 [B * Sp/64] (as int32 tensor): the 128-row tiles of G^T each 64-row k-step can touch (include/fbl.h)"""
    mask = torch.empty(B * (Sp // 64), dtype=torch.int32, device=relidx.device)
    _chk(load().fbl_gt_tilemask(_p(relidx), _p(klen), B, S, Sp, span2, int(neg), int(rmin), int(rcnt), _p(mask), _stream()),
         "fbl_gt_tilemask")
    return mask


def disent_attn_bwd_shear(neg, X, YT, PT, relidx, out, GT, B, S, Sp, nh, span2, y_head_major=True, klen=None,
                          rmin=0, rcnt=None, lin=0, border=None, row0=None, tilemask=None):
    ldout = _rows2d(out, "out")
    sh, sb, sd = head_strides(B, Sp, nh, y_head_major)
    _chk(load().fbl_disent_attn_bwd_shear(int(neg), _p(X), _p(YT), sh, sb, sd, _p(PT), _p(relidx), _p(klen), _p(border), _p(out), ldout,
                                          _p(GT), rmin, span2 if rcnt is None else rcnt, int(lin), B, S, Sp, nh, span2,
                                          _row0(row0, B, klen), _p(tilemask), _stream()),
         "fbl_disent_attn_bwd_shear")


def ce_fwd(logits, labels, V, row_lse, loss_sum_cnt):
    ldv = _rows2d(logits, "logits")
    _req(logits, torch.float32, "logits"); _req(labels, torch.int64, "labels")
    _chk(load().fbl_ce_fwd(_p(logits), ldv, _p(labels), logits.shape[0], V, _p(row_lse), _p(loss_sum_cnt), _stream()),
         "fbl_ce_fwd")


def ce_bwd_rows(logits, labels, rows, V, Vp, row_lse, loss_sum_cnt, gscale, dlogits):
    """gscale: python float, or a 1-element fp32 device tensor (read by the kernel: no host sync)"""
    ldv = _rows2d(logits, "logits")
    _req(rows, torch.int32, "rows")
    gdev = None
    if isinstance(gscale, torch.Tensor):
        gdev = gscale.reshape(1)
        _req(gdev, torch.float32, "gscale")
        gscale = 1.0
    _chk(load().fbl_ce_bwd_rows(_p(logits), ldv, _p(labels), _p(rows), rows.numel(), V, Vp, _p(row_lse),
                                _p(loss_sum_cnt), float(gscale), _p(gdev), _p(dlogits), _stream()), "fbl_ce_bwd_rows")


def gather_rows_bf16(x, rows, out):
    ld = _rows2d(x, "x")
    _chk(load().fbl_gather_rows_bf16(_p(x), ld, _p(rows), rows.numel(), x.shape[1], _p(out), _stream()),
         "fbl_gather_rows_bf16")


def scatter_rows_f32(x, rows, out):
    ld = _rows2d(out, "out")
    _chk(load().fbl_scatter_rows_f32(_p(x), _p(rows), rows.numel(), x.shape[1], _p(out), ld, _stream()),
         "fbl_scatter_rows_f32")


_SUMSQ_WS = {}


def sumsq(x, out, ws=None):
    """out[0] += sum x^2 (deterministic: per-block partials in `ws`, a per-device buffer of this binding by default)"""
    if ws is None:
        ws = _SUMSQ_WS.get(x.device)
        if ws is None:
            ws = _SUMSQ_WS[x.device] = torch.empty(load().fbl_sumsq_ws_floats(), dtype=torch.float32, device=x.device)
    _chk(load().fbl_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), "fbl_sumsq")


def adam_flat(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, sumsq_t=None, max_norm=0.0, grad_scale=1.0):
    _chk(load().fbl_adam_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                              float(weight_decay), int(step), _p(sumsq_t), float(max_norm), float(grad_scale),
                              _stream()), "fbl_adam_flat")


def cast_bf16(x, out):
    _req(x, torch.float32, "x"); _req(out, torch.bfloat16, "out")
    assert x.is_contiguous() and out.is_contiguous()
    _chk(load().fbl_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "fbl_cast_f32_to_bf16")


def dropout_f32(x, p_drop, seed, out_f32=None, out_bf16=None):
    _req(x, torch.float32, "x")
    assert x.is_contiguous()
    _chk(load().fbl_dropout_f32(_p(x), float(p_drop), int(seed), _seed_dev(), _p(out_f32), _p(out_bf16), x.numel(), _stream()),
         "fbl_dropout_f32")


def zero_(t):
    """t[...] = 0 on the current stream through the library (contiguous tensors only); returns t"""
    assert t.is_contiguous()
    _chk(load().fbl_zero(_p(t), t.numel() * t.element_size(), _stream()), "fbl_zero")
    return t


def zeros(*shape, dtype, device):
    return zero_(torch.empty(*shape, dtype=dtype, device=device))


def heads_to_rows_bf16(src, dst):
    """dst[e, r, h*64 + c] = src[e, h, r, c]; src fp32 [E, nh, rows, 64] contiguous, dst bf16 [E, rows, >= nh*64] (column view ok)"""
    E, nh, rows, c = src.shape
    assert c == 64 and src.is_contiguous() and dst.shape[0] == E and dst.shape[1] == rows and dst.stride(2) == 1
    assert dst.stride(0) == rows * dst.stride(1)
    _req(src, torch.float32, "src")
    _req(dst, torch.bfloat16, "dst")
    _chk(load().fbl_heads_to_rows_bf16(_p(src), _p(dst), E, nh, rows, dst.stride(1), _stream()), "fbl_heads_to_rows_bf16")


def dropout_sum_f32(x, seeds, p_drop, out, key0=0):
    """out[i] = sum_s dropout_{seeds[s]}(x[s, i]), element i keyed by key0 + i; x: [n_slices, ...] fp32 contiguous, out: one
    slice's shape (contiguous)"""
    ns = x.shape[0]
    n = x.numel() // ns
    assert x.is_contiguous() and out.is_contiguous() and out.numel() == n and len(seeds) == ns
    arr = (C.c_uint64 * ns)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds])
    _chk(load().fbl_dropout_sum_f32(_p(x), n, int(key0), ns, arr, float(p_drop), _seed_dev(), _p(out), _stream()), "fbl_dropout_sum_f32")


def dropout_bf16_(x, p_drop, seed):
    _req(x, torch.bfloat16, "x")
    assert x.is_contiguous()
    _chk(load().fbl_dropout_bf16(_p(x), float(p_drop), int(seed), _seed_dev(), x.numel(), _stream()), "fbl_dropout_bf16")
