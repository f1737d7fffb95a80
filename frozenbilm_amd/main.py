"""train_one_epoch / evaluate with the reference's signatures and return values (main.py:24-153).

Differences from the reference are confined to what it takes to keep an MI355X busy: when the optimizer is the fused
one, clip + Adam are two kernel launches on the flat trainable buffer (main.py:82-84 semantics), and under
torch.distributed the gradients of the trainable set are all-reduced by ``parallel.GradReducer`` (attached by the
caller) -- the reference trains independent replicas.
"""
from __future__ import annotations

import torch

from .loops import EpochRunner, LossLog, frozen_weights, logged_step, tokenize, video_inputs
from .util.misc import mask_tokens


def _prepare(batch_dict, tokenizer, device, args, step_seed=0):
    """main.py:41-58.  Two opt-in device-side shortcuts that leave the reference path untouched: a batch already staged
    on the GPU by ``datasets.stage_packed_batch`` brings its own ``video_mask``; ``args.device_mask_tokens`` runs the MLM
    corruption as one kernel on the GPU (same distribution; the default host path reproduces the reference's RNG)."""
    video, video_mask = video_inputs(batch_dict, device)
    encoded = tokenize(tokenizer, batch_dict["text"], args)
    if getattr(args, "device_mask_tokens", False):
        from .util.misc import mask_tokens_device

        inputs, labels = mask_tokens_device(encoded["input_ids"].to(device), tokenizer, args.mlm_prob, seed=step_seed)
    else:
        inputs, labels = mask_tokens(encoded["input_ids"], tokenizer, mlm_probability=args.mlm_prob)
    return dict(video=video, video_mask=video_mask, input_ids=inputs.to(device),
                attention_mask=encoded["attention_mask"].to(device), labels=labels.to(device))


def train_one_epoch(model, tokenizer, data_loader, optimizer, device, epoch, args, max_norm):
    model.train()
    if getattr(args, "packed_rows", False) and hasattr(model, "packed_rows"):
        model.packed_rows = True  # opt-in: ragged batches without the padding rows behind each sample's last token
    run = EpochRunner(data_loader, args, "Epoch: [{}]".format(epoch), epoch)
    log = LossLog(run, "mlm_loss", delayed=getattr(args, "delayed_loss_check", False), reducer=getattr(model, "_reducer", None))
    for i_batch, batch_dict in run:
        feed = _prepare(batch_dict, tokenizer, device, args, step_seed=run.global_step(i_batch) + 1)
        loss = model(**feed)["loss"]
        logged_step(log, loss, optimizer, model, max_norm)  # main.py:70-86 (see loops.LossLog for the order of the loss read)
        run.schedule(optimizer, i_batch)
    return run.finish()


@torch.no_grad()
def evaluate(model, tokenizer, data_loader, device, args):
    model.eval()
    if getattr(args, "packed_rows", False) and hasattr(model, "packed_rows"):
        model.packed_rows = True
    run = EpochRunner(data_loader, args, "Val:")
    log = LossLog(run, "mlm_loss", stop_on_nonfinite=False, delayed=True)  # (nothing to protect: read one batch late)
    with frozen_weights(model):
        for _, batch_dict in run:
            log(model(**_prepare(batch_dict, tokenizer, device, args))["loss"])
    return run.finish(synchronize=False)
