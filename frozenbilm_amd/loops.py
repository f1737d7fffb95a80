"""Pieces shared by the three training / evaluation loop modules (main.py, videoqa.py, mc.py).

The reference repeats this logic in each script (main.py:60-95, videoqa.py:62-112, mc.py:94-127); here it lives once:
tokenise, reduce-and-check the logged loss, and apply one optimizer step -- fused clip + Adam when the optimizer is
`FusedAdam`, `clip_grad_norm_` + `step()` otherwise, with the data-parallel exchange held until the last backward pass
when several forwards feed one step.
"""
from __future__ import annotations

import contextlib
import math
import sys

import torch

from .optim import FusedAdam
from .util import dist
from .util.metrics import MetricLogger
from .util.misc import adjust_learning_rate, get_mask


def tokenize(tokenizer, text, args):
    return tokenizer(text, add_special_tokens=True, max_length=args.max_tokens, padding="longest", truncation=True,
                     return_tensors="pt")


def video_inputs(batch_dict, device):
    """(video, video_mask) on the device; a batch staged by `datasets.stage_packed_batch` brings its own mask"""
    video = batch_dict["video"].to(device)
    if "video_mask" in batch_dict:
        return video, batch_dict["video_mask"].to(device)
    return video, get_mask(batch_dict["video_len"], video.size(1)).to(device)


class LossLog:
    """The per-step loss bookkeeping of the training / evaluation loops (main.py:70-78, videoqa.py:84-95, mc.py:94-105):
    all-reduce the loss dict for logging, stop the run on a non-finite training loss, feed the meters.

    The reference reads the loss on the host (`.item()`) BEFORE it calls backward: the host waits for the forward in flight
    and the GPU then idles while the host enqueues the ~700 launches of the backward (22 % of the step on one MI355X).
    Same observable behaviour without the idle time -- the default of the training loops here:

        log.begin(loss)                      # stage the loss: an asynchronous copy to pinned host memory behind the forward
        optimizer_step(..., check=log.check) # zero_grad, ENQUEUE backward, then read the loss: non-finite -> message and
                                             # sys.exit(1) before optimizer.step() -- no update is ever applied after a
                                             # non-finite loss, exactly as main.py:73-84; finite -> meters, then the step

    Under data parallelism with a `GradReducer` the loss rides in front of the first gradient bucket (parallel.py: no
    collective and no synchronisation of its own); without one, `dist.reduce_dict` as in the reference.  `log(loss)` keeps
    the reference's literal order (read, then backward) for callers that want it, and ``args.delayed_loss_check`` (opt-in)
    moves the read to the NEXT step's call (evaluation loops do that by default: they have no update to protect)."""

    def __init__(self, run: "EpochRunner", name: str, stop_on_nonfinite: bool = True, delayed: bool = False, reducer=None):
        self.run, self.name, self.stop, self.delayed = run, name, stop_on_nonfinite, delayed
        self.reducer = reducer if (reducer is not None and getattr(reducer, "carries_scalars", False)) else None
        self.pending = None
        self._staged_loss = None
        self._host = None
        self._flip = 0
        run.loss_log = self

    def _emit(self, reduced_host):
        value = float(sum(reduced_host.values()))
        if self.stop and not math.isfinite(value):
            print("Loss is {}, stopping training".format(value))
            print(reduced_host)
            sys.exit(1)
        self.run.log(loss=value, **reduced_host)

    def _stage(self, keys, values):
        """asynchronous copy of the (already reduced) scalars to pinned memory; two alternating buffers"""
        if self._host is None:
            self._host = [torch.empty(8, dtype=torch.float32).pin_memory() for _ in range(2)]
        host = self._host[self._flip]
        self._flip ^= 1
        host[: len(keys)].copy_(torch.stack([v.detach().float().reshape(()) for v in values]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = ("host", keys, host, ev)

    def __call__(self, loss):
        """the reference's order: reduce, read on the host now (or, delayed, when the next step calls)"""
        reduced = dist.reduce_dict({self.name: loss})
        if not (self.delayed and all(torch.is_tensor(v) and v.is_cuda for v in reduced.values())):
            self._emit({k: (v.item() if torch.is_tensor(v) else float(v)) for k, v in reduced.items()})
            return
        self.flush()
        keys = list(reduced)
        self._stage(keys, [reduced[k] for k in keys])

    def begin(self, loss):
        """stage the loss of this step; `check()` reads it after the backward has been enqueued"""
        self.flush()
        if not (torch.is_tensor(loss) and loss.is_cuda):
            self(loss)  # host-side losses (test doubles): nothing to overlap
            return
        if self.reducer is not None and dist.get_world_size() > 1:
            self.reducer.stage_scalars(loss.reshape(1))  # leaves with the first gradient bucket of this step's backward
            self._staged_loss = loss.detach()
            self.pending = ("reducer", [self.name])
            return
        reduced = dist.reduce_dict({self.name: loss})
        keys = list(reduced)
        self._stage(keys, [reduced[k] for k in keys])

    def check(self):
        self.flush()

    def flush(self):
        if self.pending is None:
            return
        pend, self.pending = self.pending, None
        if pend[0] == "reducer":
            vals = self.reducer.take_scalars()
            if vals is None:
                # No collective of this step carried the staged loss (a backward that did not go through the engine's reducer,
                # or a reducer still held): the stop-before-update guarantee must not depend on that -- reduce and read the
                # loss now, like the path without a reducer (main.py:73-78 checks the loss before every update).
                loss = self._staged_loss
                if loss is None:
                    return
                reduced = dist.reduce_dict({self.name: loss})
                self._emit({k: float(v.item()) for k, v in reduced.items()})
                return
            self._emit(dict(zip(pend[1], vals)))
            return
        _, keys, host, ev = pend
        ev.synchronize()
        self._emit({k: float(host[i]) for i, k in enumerate(keys)})


def frozen_weights(model):
    """`model.weights_frozen()` where the model has it (the MI355X model), a null context otherwise (test doubles)"""
    return model.weights_frozen() if hasattr(model, "weights_frozen") else contextlib.nullcontext()


def optimizer_step(loss, optimizer, model, max_norm, reducer=None, check=None):
    """zero_grad -> backward -> (clip) -> step  (main.py:80-86).  check: called between the (enqueued) backward and the
    update -- `LossLog.check`, which stops the run on a non-finite loss before any parameter changes."""
    optimizer.zero_grad()
    hold = reducer.accumulate() if reducer is not None else contextlib.nullcontext()
    with hold:
        loss.backward()
    if check is not None:
        check()
    if isinstance(optimizer, FusedAdam):
        optimizer.step(clip_max_norm=max_norm)
        return
    if max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    optimizer.step()


def logged_step(log: "LossLog", loss, optimizer, model, max_norm, reducer=None):
    """One training step with the reference's loss bookkeeping (main.py:70-86, videoqa.py:84-101, mc.py:94-113): by default
    the backward is enqueued before the host reads the loss (no update after a non-finite loss, see LossLog); with the
    opt-in ``delayed_loss_check`` the loss is read when the next step calls."""
    if log.delayed:
        log(loss)
        optimizer_step(loss, optimizer, model, max_norm, reducer=reducer)
    else:
        log.begin(loss)
        optimizer_step(loss, optimizer, model, max_norm, reducer=reducer, check=log.check)


class EpochRunner:
    """Iterates a data loader with the reference's logging cadence and learning-rate schedule."""

    def __init__(self, data_loader, args, header, epoch=None):
        self.loader, self.args, self.header, self.epoch = data_loader, args, header, epoch
        self.logger = MetricLogger(delimiter="  ")
        self.total_steps = int(len(data_loader) * args.epochs) if epoch is not None else 0

    def __iter__(self):
        return enumerate(self.logger.log_every(self.loader, self.args.print_freq, self.header))

    def global_step(self, i_batch: int) -> int:
        return self.epoch * len(self.loader) + i_batch

    def schedule(self, optimizer, i_batch: int) -> None:
        adjust_learning_rate(optimizer, curr_step=self.global_step(i_batch), num_training_steps=self.total_steps, args=self.args)

    def log(self, **scalars) -> None:
        self.logger.update(**scalars)

    def finish(self, synchronize=True):
        if getattr(self, "loss_log", None) is not None:
            self.loss_log.flush()
        if synchronize:
            self.logger.synchronize_between_processes()
            print("Averaged stats:", self.logger)
        return {name: meter.global_avg for name, meter in self.logger.meters.items()}
