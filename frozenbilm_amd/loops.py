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


def logged_loss(name: str, loss, stop_on_nonfinite: bool = True):
    """All-reduced loss dict for logging + its scalar; in training a non-finite loss stops the run (main.py:70-78)."""
    reduced = dist.reduce_dict({name: loss})
    value = sum(reduced.values()).item()
    if stop_on_nonfinite and not math.isfinite(value):
        print("Loss is {}, stopping training".format(value))
        print(reduced)
        sys.exit(1)
    return reduced, value


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
        if synchronize:
            self.logger.synchronize_between_processes()
            print("Averaged stats:", self.logger)
        return {name: meter.global_avg for name, meter in self.logger.meters.items()}
