"""Relative-position bucket tables (host side, integer-exact).

DeBERTa-v2 maps delta = i - j to a log bucket (reference: model/deberta.py:578-618, float64 log + ceil) and the
attention bias indexes the 2*span position table at clamp(bucket + span, 0, 2*span - 1) (:873, :897).  Both the c2p
and the p2c index are the same Toeplitz function of delta (bucket is odd), so one int16 vector of length 2S-1 is all
the HIP kernel needs.  Tables are cached per (S, buckets, max_pos): the reference rebuilds them with numpy three
times per forward.
"""
from __future__ import annotations

import functools

import numpy as np


def bucket_of_delta(delta: np.ndarray, bucket_size: int, max_position: int) -> np.ndarray:
    d = np.asarray(delta, dtype=np.int64)
    if bucket_size <= 0 or max_position <= 0:
        return d
    half = bucket_size // 2
    mag = np.abs(d)
    far = mag >= half  # |delta| < half keeps its identity bucket
    safe = np.where(far, mag, half).astype(np.float64)
    logb = np.ceil(np.log(safe / half) / np.log((max_position - 1) / half) * (half - 1)).astype(np.int64) + half
    return np.where(far & (mag > half), np.sign(d) * logb, d)


@functools.lru_cache(maxsize=64)
def rel_index_vector(S: int, bucket_size: int, max_position: int, span: int) -> np.ndarray:
    """int16 [2S-1]: out[d + S - 1] = clamp(bucket(d) + span, 0, 2*span - 1)."""
    d = np.arange(-(S - 1), S, dtype=np.int64)
    idx = np.clip(bucket_of_delta(d, bucket_size, max_position) + span, 0, 2 * span - 1)
    assert (np.diff(idx) >= 0).all() and (np.diff(idx) <= 1).all(), "index must be monotone with slope <= 1"
    return idx.astype(np.int16)
