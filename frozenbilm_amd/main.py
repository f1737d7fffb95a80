"""train_one_epoch / evaluate with the reference's signatures and return values (main.py:24-153).

Differences from the reference are confined to what it takes to keep an MI355X busy: when the optimizer is the fused
one, clip + Adam are two kernel launches on the flat trainable buffer (main.py:82-84 semantics), and under
torch.distributed the gradients of the trainable set are all-reduced by ``parallel.GradReducer`` (attached by the
caller) -- the reference trains independent replicas.
"""
from __future__ import annotations

import math
import sys

import torch

from .optim import FusedAdam
from .util import dist
from .util.metrics import MetricLogger
from .util.misc import adjust_learning_rate, get_mask, mask_tokens


def _prepare(batch_dict, tokenizer, device, args, step_seed=0):
    """main.py:41-58.  Two opt-in device-side shortcuts that leave the reference path untouched: a batch already staged
    on the GPU by ``datasets.stage_packed_batch`` brings its own ``video_mask``; ``args.device_mask_tokens`` runs the MLM
    corruption as one kernel on the GPU (same distribution; the default host path reproduces the reference's RNG)."""
    video = batch_dict["video"].to(device)
    video_len = batch_dict["video_len"]
    video_mask = batch_dict["video_mask"].to(device) if "video_mask" in batch_dict else get_mask(video_len, video.size(1)).to(device)
    encoded = tokenizer(batch_dict["text"], add_special_tokens=True, max_length=args.max_tokens, padding="longest",
                        truncation=True, return_tensors="pt")
    if getattr(args, "device_mask_tokens", False):
        from .util.misc import mask_tokens_device

        inputs, labels = mask_tokens_device(encoded["input_ids"].to(device), tokenizer, args.mlm_prob, seed=step_seed)
    else:
        inputs, labels = mask_tokens(encoded["input_ids"], tokenizer, mlm_probability=args.mlm_prob)
    return dict(video=video, video_mask=video_mask, input_ids=inputs.to(device),
                attention_mask=encoded["attention_mask"].to(device), labels=labels.to(device))


def train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, args, max_norm):
    model.train()
    metric_logger = MetricLogger(delimiter="  ")
    header = "Epoch: [{}]".format(epoch)
    num_training_steps = int(len(data_loader) * args.epochs)
    for i_batch, batch_dict in enumerate(metric_logger.log_every(data_loader, args.print_freq, header)):
        feed = _prepare(batch_dict, tokenizer, device, args, step_seed=epoch * len(data_loader) + i_batch + 1)
        output = model(**feed)
        loss = output["loss"]
        loss_dict_reduced = dist.reduce_dict({"mlm_loss": loss})
        loss_value = sum(loss_dict_reduced.values()).item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            print(loss_dict_reduced)
            sys.exit(1)
        optimizer.zero_grad()
        loss.backward()
        if isinstance(optimizer, FusedAdam):
            optimizer.step(clip_max_norm=max_norm)
        else:
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            optimizer.step()
        adjust_learning_rate(optimizer, curr_step=epoch * len(data_loader) + i_batch,
                             num_training_steps=num_training_steps, args=args)
        metric_logger.update(loss=loss_value, **loss_dict_reduced)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(model, tokenizer, data_loader, device, args):
    model.eval()
    metric_logger = MetricLogger(delimiter="  ")
    for i_batch, batch_dict in enumerate(metric_logger.log_every(data_loader, args.print_freq, "Val:")):
        feed = _prepare(batch_dict, tokenizer, device, args)
        output = model(**feed)
        loss_dict_reduced = dist.reduce_dict({"mlm_loss": output["loss"]})
        loss_value = sum(loss_dict_reduced.values()).item()
        metric_logger.update(loss=loss_value, **loss_dict_reduced)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
