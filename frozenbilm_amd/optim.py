"""Fused clip + Adam over the flat trainable buffer (two kernel launches per step instead of ~300 x k).

Drop-in for ``torch.optim.Adam(params_requiring_grad, lr, betas, weight_decay)`` as built by the reference
(main.py:182-188) followed by ``clip_grad_norm_(model.parameters(), max_norm)`` (main.py:82-84): pass
``clip_max_norm`` to ``step`` to fold the global-norm clip into the update.  ``param_groups[0]["lr"]`` is honoured so
``adjust_learning_rate`` (util/misc.py:59-78) keeps working.

One deliberate difference from ``torch.optim.Adam``: the update runs over the whole flat trainable buffer with ONE step
counter, so a parameter whose gradient is identically zero in a step (``linear_video`` on a batch without video -- no
loop of the reference produces one) still sees its moments decay and moves by ``lr * m_hat / (sqrt(v_hat) + eps)``,
whereas torch skips parameters whose ``.grad`` is None.  Exported states carry the common step count for every parameter.
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
        self._ss_ws = None

    def zero_grad(self, set_to_none: bool = True):
        eng = self.model._engine
        if eng is not None and not set_to_none:
            from . import lib as L

            L.zero_(eng.flat_grad)
            return
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, clip_max_norm: float = 0.0, grad_scale: float = 1.0):
        eng = self.model.engine()
        flat, g = eng.flat, eng.flat_grad
        if self._m is None:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._ss = torch.zeros(1, dtype=torch.float32, device=flat.device)
            self._ss_ws = torch.empty(L.load().fbl_sumsq_ws_floats(), dtype=torch.float32, device=flat.device)
        elif self._m.numel() != flat.numel():
            raise RuntimeError("the trainable set changed size under a live optimizer state")
        elif self._m.device != flat.device:  # the model moved: the moments follow it
            self._m, self._v = self._m.to(flat.device), self._v.to(flat.device)
            self._ss = torch.zeros(1, dtype=torch.float32, device=flat.device)
            self._ss_ws = torch.empty(L.load().fbl_sumsq_ws_floats(), dtype=torch.float32, device=flat.device)
        grp = self.param_groups[0]
        self._step += 1
        ss = None
        if clip_max_norm and clip_max_norm > 0:
            self._ss.zero_()
            if self._ss_ws is None or self._ss_ws.device != g.device:
                self._ss_ws = torch.empty(L.load().fbl_sumsq_ws_floats(), dtype=torch.float32, device=g.device)
            L.sumsq(g, self._ss, ws=self._ss_ws)
            ss = self._ss
        b1, b2 = grp["betas"]
        L.adam_flat(flat, g, self._m, self._v, grp["lr"], b1, b2, grp["eps"], grp["weight_decay"], self._step, sumsq_t=ss,
                    max_norm=clip_max_norm or 0.0, grad_scale=grad_scale)
        eng.params_version += 1  # the bf16 operand copies of the engine are stale now

    def grad_norm(self) -> torch.Tensor:
        """global L2 norm of the last clipped step's gradients (device scalar)"""
        return self._ss.sqrt() if self._ss is not None else torch.zeros(())

    # ---- checkpoints: torch.optim.Adam's schema, so files written by the reference (main.py:290-300 saves
    # optimizer.state_dict() of torch.optim.Adam over [p for p in model.parameters() if p.requires_grad], main.py:182-188)
    # resume here and files written here resume there.  Parameter i of the schema is the i-th trainable parameter in
    # model.parameters() order; its moments are the slice of the flat m / v buffers at that parameter's offset.
    def _slices(self, eng):
        by_id = {id(eng.named[n]): (eng.offsets[n], eng.named[n].numel(), tuple(eng.named[n].shape)) for n in eng.order}
        return [by_id[id(p)] for p in self.param_groups[0]["params"]]

    def state_dict(self):
        eng = self.model.engine()
        state = {}
        if self._m is not None and self._step > 0:
            for i, (o, k, shape) in enumerate(self._slices(eng)):
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[o:o + k].view(shape).clone(),
                            "exp_avg_sq": self._v[o:o + k].view(shape).clone()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(len(g["params"])))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        eng = self.model.engine()
        flat = eng.flat
        m = torch.zeros_like(flat)
        v = torch.zeros_like(flat)
        if "state" in sd:  # torch.optim.Adam schema (reference-written or ours)
            sl = self._slices(eng)
            steps = set()
            for i, st in sd["state"].items():
                i = int(i)
                if i >= len(sl):
                    raise ValueError(f"optimizer state for parameter {i}, but only {len(sl)} trainable parameters")
                o, k, shape = sl[i]
                if st["exp_avg"].numel() != k or st["exp_avg_sq"].numel() != k:
                    raise ValueError(f"optimizer state {i}: {tuple(st['exp_avg'].shape)} does not match parameter {shape}")
                m[o:o + k].copy_(st["exp_avg"].reshape(-1).to(flat.device, flat.dtype))
                v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1).to(flat.device, flat.dtype))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused update keeps one")
            self._step = steps.pop() if steps else 0
        else:  # flat schema of earlier builds: {"step", "m", "v"}
            if sd["m"] is not None:
                if sd["m"].numel() != flat.numel() or sd["v"].numel() != flat.numel():
                    raise ValueError("flat optimizer state does not match the trainable buffer")
                m.copy_(sd["m"].to(flat.device, flat.dtype))
                v.copy_(sd["v"].to(flat.device, flat.dtype))
            self._step = int(sd["step"])
        self._m, self._v = m, v
        self._ss = torch.zeros(1, dtype=torch.float32, device=flat.device)
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v_ for k, v_ in s_.items() if k != "params"})
