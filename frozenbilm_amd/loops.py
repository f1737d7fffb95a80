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

    The reference reads the loss on the host (`.item()`) before it calls backward, i.e. the host waits for the forward in
    flight and the GPU then waits for the host to enqueue the backward.  Default here: exactly that.  With
    ``args.delayed_loss_check`` (opt-in) the reduced loss of step i goes to pinned host memory by an asynchronous copy and
    is checked / logged when step i+1 calls -- the copy has long finished, nothing waits -- and `flush()` (called by
    `EpochRunner.finish`) handles the last step.  A non-finite loss is then noticed one step late: the run still stops
    after one more optimizer step than the reference would have applied; the returned averages are identical."""

    def __init__(self, run: "EpochRunner", name: str, stop_on_nonfinite: bool = True, delayed: bool = False):
        self.run, self.name, self.stop, self.delayed = run, name, stop_on_nonfinite, delayed
        self.pending = None
        run.loss_log = self

    def _emit(self, reduced_host):
        value = float(sum(reduced_host.values()))
        if self.stop and not math.isfinite(value):
            print("Loss is {}, stopping training".format(value))
            print(reduced_host)
            sys.exit(1)
        self.run.log(loss=value, **reduced_host)

    def __call__(self, loss):
        reduced = dist.reduce_dict({self.name: loss})
        if not (self.delayed and all(torch.is_tensor(v) and v.is_cuda for v in reduced.values())):
            self._emit({k: (v.item() if torch.is_tensor(v) else float(v)) for k, v in reduced.items()})
            return
        self.flush()
        keys = list(reduced)
        host = torch.empty(len(keys), dtype=torch.float32).pin_memory()
        host.copy_(torch.stack([reduced[k].detach().float().reshape(()) for k in keys]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (keys, host, ev)

    def flush(self):
        if self.pending is not None:
            keys, host, ev = self.pending
            self.pending = None
            ev.synchronize()
            self._emit({k: float(host[i]) for i, k in enumerate(keys)})


def frozen_weights(model):
    """`model.weights_frozen()` where the model has it (the MI355X model), a null context otherwise (test doubles)"""
    return model.weights_frozen() if hasattr(model, "weights_frozen") else contextlib.nullcontext()


def optimizer_step(loss, optimizer, model, max_norm, reducer=None):
    """zero_grad -> backward -> (clip) -> step  (main.py:80-86)."""
    optimizer.zero_grad()
    hold = reducer.accumulate() if reducer is not None else contextlib.nullcontext()
    with hold:
        loss.backward()
    if isinstance(optimizer, FusedAdam):
        optimizer.step(clip_max_norm=max_norm)
        return
    if max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    optimizer.step()


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
