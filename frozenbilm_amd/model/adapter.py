"""Bottleneck Adapter: drop-in for the reference's model/adapter.py (same ctor / forward signature).

    forward(x) = x + up(dropout(relu(down(x))))                      (model/adapter.py:33-45, default flags)

Inside DebertaV2ForMaskedLM the adapters are executed by the engine (fused into the sub-layer pipeline); this
stand-alone module runs the same C-ABI kernels for callers that use an Adapter on its own.  The ``ln_before`` /
``ln_after`` options are never enabled by the reference model (model/deberta.py:252,326) and are rejected.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import lib as L


class _AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wd, bd, wu, bu, p_drop, seed):
        shp = x.shape
        H = shp[-1]
        A = wd.shape[0]
        Ap = (A + 63) // 64 * 64
        x2 = x.reshape(-1, H).contiguous().float()
        N = x2.shape[0]
        dev = x.device
        xb = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
        L.cast_bf16(x2, xb)
        wdb = wd.detach().to(torch.bfloat16).contiguous()
        wub = torch.zeros(H, Ap, dtype=torch.bfloat16, device=dev)
        wub[:, :A] = wu.detach().to(torch.bfloat16)
        z = torch.zeros(N, Ap, dtype=torch.bfloat16, device=dev)
        L.adapter_down_fwd(xb, wdb, bd.detach().float().contiguous(), z, A=A, p_drop=p_drop, seed=seed)
        y = torch.empty(N, H, dtype=torch.float32, device=dev)
        L.gemm(z, wub, bias=bu.detach().float().contiguous(), aux=x2, aux_kind=L.AUX_ADD_F32, out_f32=y)
        ctx.save_for_backward(xb, z, wd, wu)
        ctx.p_drop = p_drop
        ctx.dims = (N, H, A, Ap)
        return y.view(shp).to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        xb, z, wd, wu = ctx.saved_tensors
        N, H, A, Ap = ctx.dims
        dev = gy.device
        Np = (N + 63) // 64 * 64
        g2 = gy.reshape(-1, H).contiguous().float()
        gb = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
        L.cast_bf16(g2, gb)
        upT = wu.detach().t().contiguous().to(torch.bfloat16)  # [A,H]
        downT = torch.zeros(H, Ap, dtype=torch.bfloat16, device=dev)
        downT[:, :A] = wd.detach().t().to(torch.bfloat16)
        dz = torch.zeros(N, Ap, dtype=torch.bfloat16, device=dev)
        inv_keep = 1.0 / (1.0 - ctx.p_drop) if ctx.p_drop > 0 else 1.0
        L.gemm(gb, upT, alpha=inv_keep, aux=z, aux_kind=L.AUX_MUL_POS_BF16, out_bf16=dz, N=A)
        dx = torch.empty(N, H, dtype=torch.float32, device=dev)
        L.gemm(dz, downT, aux=g2, aux_kind=L.AUX_ADD_F32, out_f32=dx)
        gT = torch.empty(H, Np, dtype=torch.bfloat16, device=dev)
        zT = torch.empty(Ap, Np, dtype=torch.bfloat16, device=dev)
        dzT = torch.empty(Ap, Np, dtype=torch.bfloat16, device=dev)
        xT = torch.empty(H, Np, dtype=torch.bfloat16, device=dev)
        L.transpose_to_bf16(gb, gT)
        L.transpose_to_bf16(z, zT)
        L.transpose_to_bf16(dz, dzT)
        L.transpose_to_bf16(xb, xT)
        dwu = torch.zeros(H, A, dtype=torch.float32, device=dev)
        dwd = torch.zeros(A, H, dtype=torch.float32, device=dev)
        L.gemm(gT, zT, out_f32=dwu, N=A, splitk=max(1, min(8, Np // 256)))
        L.gemm(dzT, xT, out_f32=dwd, M=A, splitk=max(1, min(8, Np // 256)))
        dbu = torch.zeros(H, dtype=torch.float32, device=dev)
        dbd = torch.zeros(A, dtype=torch.float32, device=dev)
        L.colsum(gb, dbu, L.colsum_ws(H, dev))
        L.colsum(dz, dbd, L.colsum_ws(A, dev), cols=A)
        return dx.view(gy.shape).to(gy.dtype), dwd, dbd, dwu, dbu, None, None


class Adapter(nn.Module):
    def __init__(self, ds_factor, hidden_dim, ln_after=False, ln_before=False, dropout=0.1):
        super().__init__()
        assert not hidden_dim % ds_factor
        if ln_after or ln_before:
            raise NotImplementedError("ln_before/ln_after are never enabled on the FrozenBiLM path")
        self.down = nn.Linear(hidden_dim, hidden_dim // ds_factor)
        self.up = nn.Linear(hidden_dim // ds_factor, hidden_dim)
        self.dropout = float(dropout) if dropout else 0.0
        self.apply(self.init_weights)
        self._calls = 0
        # dropout stream of this instance: torch's seed (args.seed + rank in the reference's main()) and a per-instance
        # draw, so that adapters do not share masks and data-parallel ranks differ (model/adapter.py:41 draws from the
        # per-process torch RNG)
        self._seed_base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) << 16

    def init_weights(self, m: nn.Module, std=1e-3):
        """model/adapter.py:23-31: N(0, std) clamped to +-2 std."""
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                m.weight.normal_(std=std).clamp_(-2 * std, 2 * std)
                m.bias.normal_(std=std).clamp_(-2 * std, 2 * std)

    def forward(self, hidden_states):
        p = self.dropout if self.training else 0.0
        self._calls += 1
        return _AdapterFn.apply(hidden_states, self.down.weight, self.down.bias, self.up.weight, self.up.bias, p,
                                (self._seed_base + self._calls) & 0xFFFFFFFFFFFF)
