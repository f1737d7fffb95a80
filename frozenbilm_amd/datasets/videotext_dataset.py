"""Input side of the masked-LM path: per-video CLIP feature files -> fixed [max_feats, features_dim] tensors.

Same constructor, item and batch format as the reference (datasets/videotext_dataset.py:8-61): one fp16/fp32 ``.npy`` of
shape [n_seconds, features_dim] per video id, read as float32; more than ``max_feats`` rows are subsampled uniformly with
index ``(j * n) // max_feats``, fewer are zero-padded, ``video_len`` = number of real rows; a missing or corrupt file yields
an all-zero clip with ``video_len`` 0.  ``csv`` instead of pandas (two columns are read), index arithmetic vectorised.
"""
from __future__ import annotations

import csv
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def subsample_indices(n: int, max_feats: int) -> np.ndarray:
    """row picked for output slot j when n > max_feats (videotext_dataset.py:29-32)"""
    return (np.arange(max_feats, dtype=np.int64) * n) // max_feats


class VideoText_Dataset(Dataset):
    def __init__(self, csv_path, features_path, max_feats=10, features_dim=768):
        with open(csv_path, newline="") as f:
            rows = list(csv.DictReader(f))
        self.text = [r["text"] for r in rows]
        self.video_id = [r["video_id"] for r in rows]
        self.features = features_path
        self.max_feats = max_feats
        self.features_dim = features_dim

    def __len__(self):
        return len(self.text)

    def __getitem__(self, idx):
        try:
            video = torch.from_numpy(np.load(os.path.join(self.features, str(self.video_id[idx]) + ".mp4.npy"))).float()
            n = len(video)
            if n > self.max_feats:
                video = video[torch.from_numpy(subsample_indices(n, self.max_feats))]
                video_len = self.max_feats
            elif n < self.max_feats:
                video_len = n
                video = torch.cat([video, torch.zeros(self.max_feats - n, self.features_dim)], 0)
            else:
                video_len = self.max_feats
        except Exception:  # missing video or corrupted feature file
            video = torch.zeros(self.max_feats, self.features_dim)
            video_len = 0
        return {"video": video, "video_len": video_len, "text": self.text[idx]}


def videotext_collate_fn(batch):
    return {"video": torch.stack([b["video"] for b in batch]),
            "video_len": torch.tensor([b["video_len"] for b in batch], dtype=torch.long),
            "text": [b["text"] for b in batch]}


def build_videotext_dataset(split, args):
    if split == "train":
        csv_path = args.webvid_train_csv_path
    elif split == "val":
        csv_path = args.webvid_val_csv_path
    else:
        raise NotImplementedError
    return VideoText_Dataset(csv_path=csv_path, features_path=args.webvid_features_path, max_feats=args.max_feats,
                             features_dim=args.features_dim)


class PackedVideoText_Dataset(VideoText_Dataset):
    """Same files, but items keep the raw fp16 clip: the uniform subsample / zero pad / mask are done by one kernel on the
    GPU (`stage_packed_batch`), so the host only concatenates bytes -- the per-sample Python loop of the reference is the
    first thing to saturate at the throughput the MI355X path reaches."""

    def __getitem__(self, idx):
        try:
            clip = np.load(os.path.join(self.features, str(self.video_id[idx]) + ".mp4.npy"))
            clip = np.ascontiguousarray(clip, dtype=np.float16).reshape(-1, self.features_dim)
        except Exception:  # missing video or corrupted feature file
            clip = np.zeros((0, self.features_dim), dtype=np.float16)
        return {"clip": torch.from_numpy(clip), "text": self.text[idx]}


def packed_collate_fn(batch):
    n = torch.tensor([len(b["clip"]) for b in batch], dtype=torch.int32)
    off = torch.zeros(len(batch), dtype=torch.int64)
    off[1:] = torch.cumsum(n[:-1].long(), 0)
    clips = [b["clip"] for b in batch if len(b["clip"])]
    feats = torch.cat(clips, 0) if clips else torch.zeros(1, batch[0]["clip"].shape[1], dtype=torch.float16)
    return {"feats": feats, "row_off": off, "n_rows": n, "text": [b["text"] for b in batch]}


def stage_packed_batch(batch, max_feats, device):
    """-> the reference batch format, on the device: video fp32 [B,T,F], video_len int64 [B] (+ video_mask [B,T])"""
    from .. import lib as L

    video, vlen, vmask = L.video_stage_f16(batch["feats"].to(device, non_blocking=True), batch["row_off"].to(device),
                                           batch["n_rows"].to(device), max_feats)
    return {"video": video, "video_len": vlen, "video_mask": vmask, "text": batch["text"]}
