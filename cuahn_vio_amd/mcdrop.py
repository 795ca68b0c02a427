"""numpy evaluation of the MC-dropout mask function specified in include/hnet_rng.h.

Used by the golden-vector generator (to inject the masks into the reference model in place of
``nn.Dropout``, reference model_to_trace.py:222-235) and by host-side tests.  The HIP kernels and
the C oracle evaluate the same integer function.
"""
from __future__ import annotations

import numpy as np

STREAM_MEAN_IN, STREAM_MEAN_HID, STREAM_UNC_IN, STREAM_UNC_HID = 0, 1, 2, 3
_M32 = 0xFFFFFFFF


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def pair_key(mc_seed: int, pair_seq: int) -> int:
    return (mc_seed ^ ((pair_seq * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def drop_threshold(p: float) -> int:
    t = float(np.float32(p)) * 16777216.0
    if t <= 0.0:
        return 0
    if t >= 16777216.0:
        return 16777216
    return int(t)


def mask_prefix(key: int, stream: int, samples: np.ndarray) -> np.ndarray:
    klo, khi = key & _M32, (key >> 32) & _M32
    h0 = int(_mix32(np.array([klo ^ int(_mix32(np.array([khi ^ 0x5BD1E995]))[0])]))[0])
    s = samples.astype(np.uint64)
    return _mix32((h0 + stream * 0x9E3779B9 + s * 0x85EBCA6B + 1) & _M32)


def keep_mask(mc_seed: int, pair_seq: int, stream: int, n_samples: int, n_elem: int, p: float,
              sample_offset: int = 0) -> np.ndarray:
    """bool [n_samples, n_elem]; True = element kept.  ``sample_offset`` gives the global index of
    the first sample (MC-dropout sharding keeps masks rank-invariant)."""
    key = pair_key(mc_seed, pair_seq)
    pre = mask_prefix(key, stream, np.arange(sample_offset, sample_offset + n_samples))[:, None]
    el = np.arange(n_elem, dtype=np.uint64)[None, :]
    bits = _mix32(pre ^ ((el * 0xC2B2AE35 + 0x27D4EB2F) & _M32)) >> 8
    return bits >= drop_threshold(p)


def scale(p: float) -> np.float32:
    """inverted-dropout scale 1/(1-p) in float32, as PyTorch computes it"""
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))
