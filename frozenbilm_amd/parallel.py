"""Data-parallel gradient exchange for the trainable set only (RCCL over xGMI).

NEW capability relative to the reference, which never all-reduces gradients.  All trainable
gradients live in ONE flat fp32 buffer ordered by backward completion (engine.flat_order); the backward pipeline calls
``ready(bucket)`` as soon as a stage's gradients are final and the reducer launches an asynchronous SUM all-reduce on
the contiguous slice [cursor, bucket_end) -- RCCL runs it on its own HIP stream, overlapped with the rest of backward.
``finish()`` joins the collectives and scales by 1/world (DDP convention: mean of per-rank mean losses).  Frozen
parameters never enter a bucket: 120.5 MB fp32 per step at DeBERTa-v2-XLarge, ~26 collectives of ~4.7 MB.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, bucket_ends: Dict[str, int], group=None, min_bucket_elems: int = 1 << 18):
        self.flat_grad = flat_grad
        self.bucket_ends = dict(bucket_ends)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.min_bucket = min_bucket_elems
        self.cursor = 0
        self.pending: List = []
        self.launched: List[tuple] = []
        self._held = False

    @classmethod
    def attach(cls, model, group=None, **kw) -> "GradReducer":
        """Attach to ``model`` (not to one Engine instance): the model rebuilds its engine -- new flat buffers -- after
        load_state_dict / .to() / set_answer_embeddings, which in the reference flow all happen AFTER the optimizer and
        the reducer are created (main.py:182 vs :235, videoqa.py:381); ``model.engine()`` re-binds the reducer then."""
        eng = model.engine()
        red = cls(eng.flat_grad, eng.bucket_ends, group=group, **kw)
        eng.reducer = red
        model._reducer = red
        return red

    def rebind(self, eng) -> None:
        """Point the reducer at a freshly built engine's flat gradient buffer and bucket boundaries."""
        if self.pending or self.cursor:
            raise RuntimeError("GradReducer.rebind in the middle of a gradient exchange")
        self.flat_grad = eng.flat_grad
        self.bucket_ends = dict(eng.bucket_ends)
        self.launched = []
        eng.reducer = self

    @contextlib.contextmanager
    def accumulate(self):
        """Several backward passes feed ONE optimizer step (mc.py runs one forward per answer candidate and
        back-propagates through all of them): buckets are not final until the last pass, so the exchange is held back
        and done in one go on exit."""
        self._held = True
        try:
            yield self
        finally:
            self._held = False
            self.finish()

    def ready(self, key: str):
        if self._held:
            return
        end = self.bucket_ends.get(key)
        if end is None or end <= self.cursor:
            return
        if end - self.cursor < self.min_bucket and end < self.flat_grad.numel():
            return  # coalesce tiny buckets (e.g. the 2H head LayerNorm) into the next one
        self._launch(end)

    def _launch(self, end: int):
        if self.world > 1:
            sl = self.flat_grad[self.cursor:end]
            self.pending.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((self.cursor, end))
        self.cursor = end

    def finish(self):
        if self._held:
            return
        if self.cursor < self.flat_grad.numel():
            self._launch(self.flat_grad.numel())
        for w in self.pending:
            w.wait()
        self.pending.clear()
        if self.world > 1:
            self.flat_grad.mul_(1.0 / self.world)
        self.cursor = 0
        self.last_launched, self.launched = self.launched, []
