"""Checkpoints in the reference's schema (main.py:235-243,290-300; videoqa.py:355-363,470-500):
``{"model": state_dict, "optimizer": state_dict, "epoch": int, "args": Namespace}``, loaded with ``strict=False`` so the
released files -- which carry the reference's key names, sometimes only the trained subset -- drop in.  Keys the build
does not own (``position_ids`` buffers, the tied ``lm_head.decoder``) are ignored; shape mismatches are errors.
"""
from __future__ import annotations

import torch

from . import dist


def checkpoint_dict(model, optimizer, epoch, args, trainable_only: bool = False):
    sd = model.state_dict()
    if trainable_only:
        keep = {n for n, p in model.named_parameters() if p.requires_grad}
        sd = {k: v for k, v in sd.items() if k in keep}
    ck = {"model": sd, "optimizer": optimizer.state_dict() if optimizer is not None else None, "epoch": epoch,
          "args": args}
    if hasattr(model, "step_seed"):  # position of the dropout stream (an extra key: the reference's loader ignores it)
        ck["fbl"] = {"step_seed": int(model.step_seed)}
    return ck


def save_checkpoint(model, optimizer, epoch, args, path, trainable_only: bool = False):
    """main.py:290-300 (``dist.save_on_master``)."""
    dist.save_on_master(checkpoint_dict(model, optimizer, epoch, args, trainable_only), path)


def load_checkpoint(model, path, optimizer=None, resume: bool = False, map_location="cpu"):
    """main.py:235-243.  Returns (checkpoint dict, start_epoch)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = {k: v for k, v in ckpt["model"].items() if "position_ids" not in k and "lm_head.decoder" not in k}
    own = model.state_dict()
    bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if bad:
        raise RuntimeError(f"checkpoint tensors with mismatching shapes: {bad[:5]}")
    model.load_state_dict(sd, strict=False)
    start_epoch = 0
    if resume and optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        start_epoch = ckpt["epoch"] + 1
        if hasattr(model, "step_seed") and isinstance(ckpt.get("fbl"), dict):
            model.step_seed = int(ckpt["fbl"].get("step_seed", model.step_seed))
    return ckpt, start_epoch
