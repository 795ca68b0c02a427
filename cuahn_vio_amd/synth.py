"""Synthetic 320x224 grayscale frame pairs for parity tests and the benchmark.

img1 is an integer value-noise texture (bit-reproducible on any machine); img2 is the same scene
seen through a random 4-corner homography (per-corner offsets U(-max_offset, max_offset) px), sampled
bilinearly from a larger canvas and rounded to u8, so pairs are geometrically consistent.  Only
+,-,*,/ in float64 are used for img2, no transcendental functions.

Corner order and (u, v) convention follow the reference: ul, bl, br, ur
(reference trace_pytorch_model/model_to_trace.py:79-83).
"""
from __future__ import annotations

import zlib

import numpy as np

from .weights import uniform01

IMG_H, IMG_W = 224, 320
P4 = np.array([[0.0, 0.0], [0.0, IMG_H - 1.0], [IMG_W - 1.0, IMG_H - 1.0], [IMG_W - 1.0, 0.0]])
_PAD = 48


def _hash_lattice(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    x = (ix.astype(np.uint64) * 0x9E3779B1 + iy.astype(np.uint64) * 0x85EBCA77 + (seed & 0xFFFFFFFF) * 0xC2B2AE3D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return (x & 0xFF).astype(np.int64)


def _octave(h: int, w: int, log2cell: int, seed: int) -> np.ndarray:
    c = 1 << log2cell
    # the lattice values once per lattice point (not once per pixel and corner): same integers, 4 - 1000x fewer hashes
    ly, lx = np.meshgrid(np.arange(((h - 1) >> log2cell) + 2, dtype=np.int64), np.arange(((w - 1) >> log2cell) + 2, dtype=np.int64), indexing="ij")
    lat = _hash_lattice(lx, ly, seed)
    y = np.arange(h, dtype=np.int64)[:, None]
    x = np.arange(w, dtype=np.int64)[None, :]
    x0, fx = x >> log2cell, x & (c - 1)
    y0, fy = y >> log2cell, y & (c - 1)
    top = lat[y0, x0] * (c - fx) + lat[y0, x0 + 1] * fx
    bot = lat[y0 + 1, x0] * (c - fx) + lat[y0 + 1, x0 + 1] * fx
    return (top * (c - fy) + bot * fy) >> (2 * log2cell)


def canvas(seed: int) -> np.ndarray:
    """int64 texture in [0,255], shape [IMG_H+2*PAD, IMG_W+2*PAD]"""
    h, w = IMG_H + 2 * _PAD, IMG_W + 2 * _PAD
    t = 4 * _octave(h, w, 5, seed * 4 + 1) + 3 * _octave(h, w, 3, seed * 4 + 2) + 2 * _octave(h, w, 2, seed * 4 + 3) \
        + 1 * _octave(h, w, 0, seed * 4 + 4)
    t = t // 10
    # stretch contrast to the full u8 range, integer arithmetic
    lo, hi = int(t.min()), int(t.max())
    return ((t - lo) * 255) // max(hi - lo, 1)


def dlt_h(offsets: np.ndarray) -> np.ndarray:
    """float64 4-point homography p4 -> p4 + offsets (same linear system as the reference's
    DLT_solve, model_to_trace.py:42-61), used only to *generate* data."""
    dst = P4 + np.asarray(offsets, dtype=np.float64).reshape(4, 2)
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = P4[i]
        u, v = dst[i]
        a[2 * i] = [x, y, 1, 0, 0, 0, -u * x, -u * y]
        a[2 * i + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]
        b[2 * i], b[2 * i + 1] = u, v
    h8 = np.linalg.solve(a, b)
    return np.append(h8, 1.0).reshape(3, 3)


def _bilinear(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    h, w = img.shape
    x = np.clip(x, 0.0, w - 1.001)
    y = np.clip(y, 0.0, h - 1.001)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    f = img.astype(np.float64)
    return (f[y0, x0] * (1 - fx) + f[y0, x0 + 1] * fx) * (1 - fy) + (f[y0 + 1, x0] * (1 - fx) + f[y0 + 1, x0 + 1] * fx) * fy


def true_offsets(seed: int, max_offset: float = 12.0) -> np.ndarray:
    """the 8 ground-truth corner offsets (px) of pair `seed`, order ul.u ul.v bl.u ... ur.v"""
    return ((uniform01(seed, 1001, 8).astype(np.float64) * 2.0 - 1.0) * max_offset)


def make_pair(seed: int, max_offset: float = 12.0):
    """returns (img1 u8 [224,320], img2 u8 [224,320], offsets float64 [8]) with img2(H x) ~ img1(x)"""
    cv = canvas(seed)
    img1 = cv[_PAD:_PAD + IMG_H, _PAD:_PAD + IMG_W].astype(np.uint8)
    off = true_offsets(seed, max_offset)
    hm = dlt_h(off)
    # img1 pixel x maps to img2 pixel H x  =>  img2(q) = scene(H^-1 q)
    hinv = np.linalg.inv(hm)
    vs, us = np.meshgrid(np.arange(IMG_H, dtype=np.float64), np.arange(IMG_W, dtype=np.float64), indexing="ij")
    xw = hinv[0, 0] * us + hinv[0, 1] * vs + hinv[0, 2]
    yw = hinv[1, 0] * us + hinv[1, 1] * vs + hinv[1, 2]
    zw = hinv[2, 0] * us + hinv[2, 1] * vs + hinv[2, 2]
    img2 = np.floor(_bilinear(cv, xw / zw + _PAD, yw / zw + _PAD) + 0.5)
    return img1, np.clip(img2, 0, 255).astype(np.uint8), off


def make_prior(seed: int, offsets: np.ndarray, sigma: float = 2.0) -> np.ndarray:
    """EKF-prior stand-in: true offsets + bounded pseudo-noise (uniform, +-sigma*sqrt(3)), float32 [8]"""
    n = (uniform01(seed, 2002, 8).astype(np.float64) * 2.0 - 1.0) * sigma * np.sqrt(3.0)
    return (np.asarray(offsets, dtype=np.float64) + n).astype(np.float32)


def make_noise_pair(seed: int):
    """stress pair: i.i.d. uniform u8 images (no geometric consistency)"""
    a = (uniform01(seed, 3003, IMG_H * IMG_W) * 256.0).astype(np.uint8).reshape(IMG_H, IMG_W)
    b = (uniform01(seed, 3004, IMG_H * IMG_W) * 256.0).astype(np.uint8).reshape(IMG_H, IMG_W)
    return a, b


def make_batch(first_seed: int, count: int, max_offset: float = 12.0):
    """(prev u8 [B,224,320], curr u8 [B,224,320], prior f32 [B,8], offsets f64 [B,8])"""
    prev = np.empty((count, IMG_H, IMG_W), np.uint8)
    curr = np.empty((count, IMG_H, IMG_W), np.uint8)
    prior = np.empty((count, 8), np.float32)
    offs = np.empty((count, 8), np.float64)
    for i in range(count):
        prev[i], curr[i], offs[i] = make_pair(first_seed + i, max_offset)
        prior[i] = make_prior(first_seed + i, offs[i])
    return prev, curr, prior, offs


def crc(*arrays) -> int:
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c & 0xFFFFFFFF
