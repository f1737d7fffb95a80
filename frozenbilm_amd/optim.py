"""Fused clip + Adam over the flat trainable buffer (two kernel launches per step instead of ~300 x k).

Drop-in for ``torch.optim.Adam(params_requiring_grad, lr, betas, weight_decay)`` as built by the reference
(main.py:182-188) followed by ``clip_grad_norm_(model.parameters(), max_norm)`` (main.py:82-84): pass
``clip_max_norm`` to ``step`` to fold the global-norm clip into the update.  ``param_groups[0]["lr"]`` is honoured so
``adjust_learning_rate`` (util/misc.py:59-78) keeps working.
"""
from __future__ import annotations

import torch

from . import lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._m = self._v = None
        self._step = 0
        self._ss = None

    def zero_grad(self, set_to_none: bool = True):
        eng = self.model._engine
        if eng is not None and not set_to_none:
            eng.flat_grad.zero_()
            return
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, clip_max_norm: float = 0.0, grad_scale: float = 1.0):
        eng = self.model.engine()
        flat, g = eng.flat, eng.flat_grad
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._ss = torch.zeros(1, dtype=torch.float32, device=flat.device)
        grp = self.param_groups[0]
        self._step += 1
        ss = None
        if clip_max_norm and clip_max_norm > 0:
            self._ss.zero_()
            L.sumsq(g, self._ss)
            ss = self._ss
        b1, b2 = grp["betas"]
        L.adam_flat(flat, g, self._m, self._v, grp["lr"], b1, b2, grp["eps"], grp["weight_decay"], self._step, sumsq_t=ss,
                    max_norm=clip_max_norm or 0.0, grad_scale=grad_scale)

    def grad_norm(self) -> torch.Tensor:
        """global L2 norm of the last clipped step's gradients (device scalar)"""
        return self._ss.sqrt() if self._ss is not None else torch.zeros(())

    def state_dict(self):
        return {"step": self._step, "m": self._m, "v": self._v, "param_groups": [
            {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._step, self._m, self._v = sd["step"], sd["m"], sd["v"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
