"""ctypes binding of oracle/liboracle*.so (TEST INFRASTRUCTURE — see oracle/hnet_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
N_LAYERS, N_STAT = 25, 19
LAYER_NAMES = [
    "block_1_1", "block_1_2", "block_1_3", "block_2_1", "block_2_2", "block_2_3", "block_2_4",
    "block_3_0", "block_3_1", "block_3_2", "block_3_3", "block_3_4", "block_3_5",
    "block_4_0", "block_4_1", "block_4_2", "block_4_3", "block_4_4", "block_4_5", "block_4_6",
    "fc_block_1", "fc_block_2", "fc_block_3", "fc_block_4_mean", "fc_block_4_uncertainty",
]


class Trace(C.Structure):
    _fields_ = [("layer_stats", (C.c_double * N_STAT) * N_LAYERS), ("H_part1", C.c_float * 9),
                ("dlt_dst", (C.c_float * 8) * 5), ("n_dlt", C.c_int), ("feat", C.c_float * 5120)]


def build(force: bool = False) -> None:
    """compile the oracle with gcc (oracle/Makefile)"""
    if force or not all(os.path.exists(os.path.join(_DIR, n)) for n in ("liboracle.so", "liboracle_f32.so")):
        subprocess.run(["make", "-C", _DIR, "-s"] + (["-B"] if force else []), check=True)


_libs = {}


def _lib(f32: bool):
    name = "liboracle_f32.so" if f32 else "liboracle.so"
    if name not in _libs:
        build()
        lib = C.CDLL(os.path.join(_DIR, name))
        fp, u8p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_void_p
        lib.oracle_load.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        lib.oracle_load.restype = C.c_int
        lib.oracle_free.argtypes = [vp]
        lib.oracle_set_threads.argtypes = [C.c_int]
        lib.oracle_acc_bytes.restype = C.c_int
        lib.oracle_forward.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64,
                                       fp, fp, fp, C.POINTER(Trace)]
        lib.oracle_forward.restype = C.c_int
        lib.oracle_heads.argtypes = [vp, fp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, fp, fp]
        lib.oracle_finish.argtypes = [fp, fp, C.c_int, fp, fp, fp, fp]
        lib.oracle_warp.argtypes = [fp, fp, fp]
        lib.oracle_dlt.argtypes = [fp, fp]
        lib.oracle_avgpool.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        lib.oracle_conv_lrelu.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        lib.oracle_u8_to_f32.argtypes = [u8p, C.c_size_t, fp]
        _libs[name] = lib
    return _libs[name]


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def as_f32_image(img) -> np.ndarray:
    """u8 -> float32/255 exactly as HomographyNet.cpp:141 does; float32 passes through"""
    img = np.asarray(img)
    if img.dtype == np.uint8:
        return (img.astype(np.float32) / np.float32(255.0)).reshape(224, 320)
    return np.ascontiguousarray(img, dtype=np.float32).reshape(224, 320)


class Oracle:
    """one loaded weight set.  f32=False: double-accumulating checker; f32=True: plain-fp32 CPU port."""

    def __init__(self, blob: bytes, f32: bool = False, threads: int = 0):
        self.lib = _lib(f32)
        self._h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rc = self.lib.oracle_load(C.cast(buf, C.c_void_p), len(blob), C.byref(self._h))
        if rc != 0:
            raise ValueError(f"oracle_load failed: {rc}")
        # default: a modest team — the per-layer loops are small and a 256-thread team (GPU box) is far slower
        self.lib.oracle_set_threads(threads if threads else min(16, os.cpu_count() or 1))

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.oracle_free(self._h)
            self._h = None

    def forward(self, img1, img2, prior=None, blocks_to_run=3, n_mc=16, p=0.0, mc_seed=0, pair_seq=0,
                want_err=False, want_trace=False):
        i1, i2 = as_f32_image(img1), as_f32_image(img2)
        mean = np.zeros(8, np.float32)
        cov = np.zeros((8, 8), np.float32)
        err = np.zeros((224, 320), np.float32) if want_err else None
        tr = Trace() if want_trace else None
        pr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32).reshape(8)
        rc = self.lib.oracle_forward(self._h, _f(i1), _f(i2), _f(pr) if pr is not None else None, blocks_to_run,
                                     n_mc, p, mc_seed, pair_seq, _f(mean), _f(cov),
                                     _f(err) if want_err else None, C.byref(tr) if want_trace else None)
        if rc != 0:
            raise ValueError(f"oracle_forward failed: {rc}")
        out = {"mean": mean, "cov": cov}
        if want_err:
            out["err"] = err
        if want_trace:
            out["layer_stats"] = {n: np.array(tr.layer_stats[i][:]) for i, n in enumerate(LAYER_NAMES)}
            out["H_part1"] = np.array(tr.H_part1[:], np.float32).reshape(3, 3)
            out["dlt_dst"] = np.array([list(r) for r in tr.dlt_dst][: tr.n_dlt], np.float32).reshape(-1, 4, 2)
            out["feat"] = np.array(tr.feat[:], np.float32)
        return out

    def heads(self, feat, s0, s1, p, mc_seed, pair_seq):
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(5120)
        m = np.zeros((s1 - s0, 8), np.float32)
        lv = np.zeros((s1 - s0, 8), np.float32)
        self.lib.oracle_heads(self._h, _f(feat), s0, s1, p, mc_seed, pair_seq, _f(m), _f(lv))
        return m, lv

    def finish(self, mean_s, logvar_s, h_part1):
        ms = np.ascontiguousarray(mean_s, dtype=np.float32)
        ls = np.ascontiguousarray(logvar_s, dtype=np.float32)
        h1 = np.ascontiguousarray(h_part1, dtype=np.float32).reshape(9)
        mean, cov, ht = np.zeros(8, np.float32), np.zeros((8, 8), np.float32), np.zeros(9, np.float32)
        self.lib.oracle_finish(_f(ms), _f(ls), ms.shape[0], _f(h1), _f(mean), _f(cov), _f(ht))
        return mean, cov, ht.reshape(3, 3)


def warp(img, h, f32=False):
    i = as_f32_image(img)
    hm = np.ascontiguousarray(h, dtype=np.float32).reshape(9)
    out = np.zeros((224, 320), np.float32)
    _lib(f32).oracle_warp(_f(i), _f(hm), _f(out))
    return out


def dlt(dst, f32=False):
    d = np.ascontiguousarray(dst, dtype=np.float32).reshape(8)
    h = np.zeros(9, np.float32)
    _lib(f32).oracle_dlt(_f(d), _f(h))
    return h.reshape(3, 3)


def conv_lrelu(x, w, b, stride, f32=False):
    """x [Cin,H,W], w [Cout,Cin,k,k] -> [Cout,Ho,Wo]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    p = (k - 1) // 2
    ho, wo = (h + 2 * p - k) // stride + 1, (wd + 2 * p - k) // stride + 1
    out = np.zeros((cout, ho, wo), np.float32)
    _lib(f32).oracle_conv_lrelu(_f(x), cin, h, wd, _f(w), _f(b), cout, k, stride, _f(out))
    return out


def avgpool(x, k, f32=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    c, h, w = x.shape
    out = np.zeros((c, h // k, w // k), np.float32)
    _lib(f32).oracle_avgpool(_f(x), c, h, w, k, _f(out))
    return out
