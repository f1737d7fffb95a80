"""Data-parallel gradient exchange for the trainable set only (RCCL over xGMI).

NEW capability relative to the reference, which never all-reduces gradients.  All trainable
gradients live in ONE flat fp32 buffer ordered by backward completion (engine.flat_order); the backward pipeline calls
``ready(bucket)`` as soon as a stage's gradients are final and the reducer all-reduces (asynchronous SUM, RCCL's own HIP
stream) contiguous runs of ready stages; ``finish()`` joins the collectives and scales by 1/world (DDP convention: mean of
per-rank mean losses).  Frozen parameters never enter a bucket: 120.5 MB fp32 per step at DeBERTa-v2-XLarge.

WHERE in backward the collectives are launched is a knob (``overlap``), because of what the single-GPU measurements of this
repo say about anything that holds CUs next to the large GEMMs: the 8-phase GEMM runs ONE workgroup
per CU with all of its registers and LDS, so a tile cannot start on a CU on which a long-lived foreign workgroup sits -- and
an RCCL ring kernel is exactly that, one workgroup per channel for the whole transfer.

* ``"attention_windows"`` (default): ready stages are held until the engine announces a stretch of backward without such
  tiles -- the attention backward of the next layer execution (~0.35 ms of latency-bound kernels with small workgroups that
  share CUs gracefully) -- via ``window()``.  Overlapped with backward, as BASELINE's north star asks, but not on top of the
  GEMMs.
* ``"backward"``: launch at ``ready()``, whatever runs next (the round-1..3 behaviour).
* ``"after"``: one collective after the last stage (``finish()``): no overlap, no interference; 120 MB is < 1 ms on xGMI.

Stages may become ready out of order (the adapter weight gradients of a stage are final only when the group launch that
carries them has been enqueued, and the repeated last layer waits one group longer): any maximal run of adjacent ready
stages can leave, tiny ones (the 2H head LayerNorm) ride with a neighbour.  No multi-GPU node was available to this build: the
default is chosen by the reasoning above, not by measurement; ``tools/scale_sweep.sh`` (bench.py --dp-overlap ...) prints the
table that settles it.

The logged loss rides along (the reference all-reduces it on its own, util/dist.py:89-113 via
main.py:71-73): the flat gradient buffer is preceded by ``SCALAR_SLOT`` floats, ``stage_scalars`` parks the step's loss
scalar(s) there and the collective that carries the first bucket carries them too -- data parallelism adds no collective
and no host synchronisation of its own to a step.  ``take_scalars`` hands the averaged values to the host as soon as THAT
collective has finished (a copy to pinned memory on a side stream; the main stream never waits for it).
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

OVERLAP_MODES = ("attention_windows", "backward", "after")
SCALAR_SLOT = 8  # floats in front of the flat gradient buffer (32 bytes: the gradient views keep their alignment)


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, bucket_ends: Dict[str, int], group=None, min_bucket_elems: int = 1 << 18,
                 overlap: Optional[str] = None, flat_full: Optional[torch.Tensor] = None):
        self.flat_grad = flat_grad
        # flat_full: [SCALAR_SLOT + n] buffer whose tail IS flat_grad (engine.flat_grad_full), or None: no scalar slot
        self._bind_full(flat_full)
        self._staged = None       # scalars waiting for the collective that starts at offset 0
        self._scalars = None      # (k, host tensor, event | None) of this step's reduced scalars
        self._scalar_stream = None
        self.n_collectives = 0    # collectives issued since construction (tests: the loss adds none)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.min_bucket = min_bucket_elems
        self.overlap = overlap or "attention_windows"
        if self.overlap not in OVERLAP_MODES:
            raise ValueError(f"overlap must be one of {OVERLAP_MODES}, got {self.overlap!r}")
        self._set_buckets(bucket_ends)
        self.pending: List = []
        self.launched: List[Tuple[int, int]] = []
        self.last_launched: List[Tuple[int, int]] = []
        self._held = False

    def _set_buckets(self, bucket_ends: Dict[str, int]):
        self.bucket_ends = dict(bucket_ends)
        self.spans: Dict[str, Tuple[int, int]] = {}
        lo = 0
        for key, end in sorted(self.bucket_ends.items(), key=lambda kv: kv[1]):
            self.spans[key] = (lo, end)
            lo = end
        self._ready: List[Tuple[int, int]] = []  # ready, not yet launched, sorted, non-overlapping
        self._done: List[Tuple[int, int]] = []   # launched this step

    def _bind_full(self, flat_full):
        self.flat_full = None
        if flat_full is not None:
            if (flat_full.numel() != self.flat_grad.numel() + SCALAR_SLOT
                    or flat_full.data_ptr() + 4 * SCALAR_SLOT != self.flat_grad.data_ptr()):
                raise ValueError("flat_full must be the flat gradient buffer preceded by SCALAR_SLOT floats")
            self.flat_full = flat_full

    @classmethod
    def attach(cls, model, group=None, **kw) -> "GradReducer":
        """Attach to ``model`` (not to one Engine instance): the model rebuilds its engine -- new flat buffers -- after
        load_state_dict / .to() / set_answer_embeddings, which in the reference flow all happen AFTER the optimizer and
        the reducer are created (main.py:182 vs :235, videoqa.py:381); ``model.engine()`` re-binds the reducer then."""
        eng = model.engine()
        red = cls(eng.flat_grad, eng.bucket_ends, group=group, flat_full=getattr(eng, "flat_grad_full", None), **kw)
        eng.reducer = red
        model._reducer = red
        return red

    def rebind(self, eng) -> None:
        """Point the reducer at a freshly built engine's flat gradient buffer and bucket boundaries."""
        if self.pending or self._ready or self._done:
            raise RuntimeError("GradReducer.rebind in the middle of a gradient exchange")
        self.flat_grad = eng.flat_grad
        self._bind_full(getattr(eng, "flat_grad_full", None))
        self._set_buckets(eng.bucket_ends)
        self.launched = []
        eng.reducer = self

    @property
    def rccl_ranks(self) -> int:
        """ranks the collectives of this reducer run over when the backend is nccl (= RCCL on ROCm), else 0"""
        if not dist.is_initialized() or dist.get_backend(self.group) != "nccl":
            return 0
        return dist.get_world_size(self.group)

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

    # ------------------------------------------------------------------ the step's logged scalars (loss)
    @property
    def carries_scalars(self) -> bool:
        return self.flat_full is not None

    def stage_scalars(self, values: torch.Tensor) -> None:
        """Park up to SCALAR_SLOT scalars (the loss dict of the step, util/dist.py:89-113) for the exchange of THIS step: they
        are written in front of the first bucket right before its collective is launched and averaged with it."""
        v = values.detach().to(torch.float32).reshape(-1)
        if self.flat_full is None or v.numel() > SCALAR_SLOT or v.device != self.flat_full.device:
            raise ValueError("no scalar slot for these values")
        self._staged = v
        self._scalars = None

    def take_scalars(self) -> Optional[List[float]]:
        """The averaged scalars of the step, on the host: waits for the collective that carried them (not for the rest of
        backward, not for the other collectives).  None if nothing was staged or its collective has not been launched yet."""
        if self._scalars is None:
            return None
        k, host, ev = self._scalars
        self._scalars = None
        if ev is not None:
            ev.synchronize()
        return [float(x) / self.world for x in host[:k]]

    def _fetch_scalars(self, k: int, work) -> None:
        full = self.flat_full
        if not full.is_cuda:
            if work is not None:
                work.wait()
            self._scalars = (k, full[:k].clone(), None)
            return
        if self._scalar_stream is None:
            self._scalar_stream = torch.cuda.Stream(device=full.device)
            self._scalar_host = torch.empty(SCALAR_SLOT, dtype=torch.float32).pin_memory()
        st = self._scalar_stream
        st.wait_stream(torch.cuda.current_stream())  # (world 1: the write into the slot; with a collective: its launch)
        with torch.cuda.stream(st):
            if work is not None:
                work.wait()  # stream-side wait on RCCL's stream (nccl) / host wait (gloo); never on the main stream
            self._scalar_host[:k].copy_(full[:k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        self._scalars = (k, self._scalar_host, ev)

    # ------------------------------------------------------------------ called by the backward pipeline
    def ready(self, key: str):
        """the gradients of stage `key` are final (any order)"""
        if self._held:
            return
        span = self.spans.get(key)
        if span is None or span[1] <= span[0]:
            return
        self._mark(span)
        if self.overlap == "backward":
            self._launch_ready(force=False)

    def window(self):
        """a stretch of backward without one-workgroup-per-CU GEMM tiles begins (engine: attention backward)"""
        if not self._held and self.overlap == "attention_windows":
            self._launch_ready(force=False)

    # ------------------------------------------------------------------ internals
    def _mark(self, span):
        lo, hi = span
        for a, b in self._ready + self._done:
            if lo < b and a < hi:
                return  # already signalled this step
        self._ready.append((lo, hi))
        self._ready.sort()
        merged = [self._ready[0]]
        for a, b in self._ready[1:]:
            if a == merged[-1][1]:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        self._ready = merged

    def _launch_ready(self, force: bool):
        keep = []
        for lo, hi in self._ready:
            if not force and hi - lo < self.min_bucket and hi < self.flat_grad.numel():
                keep.append((lo, hi))  # tiny run: waits for a neighbour (or for finish)
                continue
            self._launch(lo, hi)
        self._ready = keep

    def _launch(self, lo: int, hi: int):
        buf, k = self.flat_grad[lo:hi], 0
        if lo == 0 and self._staged is not None:  # the collective of the first bucket carries the step's scalars
            k = self._staged.numel()
            self.flat_full[:k].copy_(self._staged)
            self._staged = None
            buf = self.flat_full[: SCALAR_SLOT + hi]  # the whole slot: the range stays 32-byte aligned
        work = None
        if self.world > 1:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append(work)
            self.n_collectives += 1
        if k:
            self._fetch_scalars(k, work)
        self.launched.append((lo, hi))
        self._done.append((lo, hi))

    def finish(self):
        if self._held:
            return
        # whatever was never signalled (or was held back) leaves now, as maximal contiguous runs
        n = self.flat_grad.numel()
        covered = sorted(self._done + self._ready)
        cur, gaps = 0, []
        for a, b in covered:
            if a > cur:
                gaps.append((cur, a))
            cur = max(cur, b)
        if cur < n:
            gaps.append((cur, n))
        for g in gaps:
            self._mark(g)
        self._launch_ready(force=True)
        for w in self.pending:
            w.wait()
        self.pending.clear()
        if self.world > 1:
            self.flat_grad.mul_(1.0 / self.world)
        self.last_launched, self.launched = self.launched, []
        self._done = []
