"""Host-side mirror of the reference runtime class ``pytorch::HomographyNet``
(reference cuahn_ros/homography_network/src/HomographyNet.h:23-67, HomographyNet.cpp) and a batched engine,
both thin wrappers over the C ABI of include/hnet.h (libhnet_hip.so).  No torch on this path.

``HomographyNet`` keeps the reference's method names, argument meaning and "print and carry on" error
behaviour so the parity tests read like a port of the reference's call sites
(cuahn/src/core/VioManager.cpp:188,236,258-259).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import PIX_F32, PIX_U8, Config, HnetError, Timing, check, lib  # noqa: F401

HIP_STREAM_LEGACY = 1      # hipStreamLegacy ((hipStream_t)1, hip_runtime_api.h): the NULL / default stream, named explicitly

IMG_H, IMG_W = 224, 320
VARIANTS = {"full": (0, 3), "prior3": (1, 3), "prior2": (1, 2), "prior1": (1, 1)}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


IGNORE_ENV = False      # bench.py sets this (unless --honour-env): engines are then built from explicit arguments only, a stray HNET_* variable cannot change the timed kernels


def env_overrides():
    """the HNET_* variables of this process that kernel_selection_from_env / HnetEngine would map onto hnet_config (what bench.py lists in `env_overrides`)"""
    names = ("HNET_S3_TILE", "HNET_FUSE_SMALL", "HNET_FUSE_B3", "HNET_FUSE_B42", "HNET_CHAIN", "HNET_CHAIN_GRID", "HNET_CHAIN_FC", "HNET_WARP_FUSE", "HNET_GRAPH_COPIES", "HNET_GRAPH", "HNET_WARP_EXACT", "HNET_PRECISION")
    return {n: os.environ[n] for n in names if n in os.environ}


def kernel_selection_from_env():
    """hnet_config.warp_exact / .graph / .variant from this process's environment.  The C library reads no environment variable (round 4);
    the test suite and tools/ab_bench.py keep selecting reference kernels with HNET_WARP_EXACT, HNET_GRAPH (0 eager, 1 replay also in the timing
    entry point), HNET_S3_TILE (13 / 20 / 21 / 22 / 25 / 30: include/hnet.h HNET_VARIANT_*), HNET_FUSE_SMALL=0, HNET_FUSE_B3=0, HNET_FUSE_B42=0 - mapped here."""
    if IGNORE_ENV:
        return 0, 0, 0
    env = os.environ.get
    variant = int(env("HNET_S3_TILE", "0")) & 0xff
    if env("HNET_FUSE_SMALL", "1") == "0":
        variant |= 1 << 8
    if env("HNET_FUSE_B3", "1") == "0":
        variant |= 1 << 9
    if env("HNET_FUSE_B42", "1") == "0":
        variant |= 1 << 10
    if env("HNET_CHAIN", "1") == "0":       # the per-layer launches instead of the one-XCD tail chains of the latency path (include/hnet.h HNET_VARIANT_NO_CHAIN)
        variant |= 1 << 11
    if env("HNET_CHAIN_GRID", "") in ("8", "3"):      # tests: the chain launches with 8 / 3 workgroups (HNET_VARIANT_CHAIN_GRID_*)
        variant |= (1 << 12) if env("HNET_CHAIN_GRID") == "8" else (1 << 13)
    if env("HNET_CHAIN_FC", "1") == "0":    # the block-tail FC recomputed by the next warp + pool launch instead of summed from the chain's partial sums (HNET_VARIANT_CHAIN_NO_FC)
        variant |= 1 << 16
    if env("HNET_WARP_FUSE", "0") == "1":   # opt-in: block 4's warp + concat sampled inside the block_4_0 + block_4_1 kernel (include/hnet.h HNET_VARIANT_WARP_FUSE)
        variant |= 1 << 14
    if env("HNET_GRAPH_COPIES", "0") == "1":   # hnet_infer's graph with memcpy nodes instead of kernels that read / write the pinned host block (HNET_VARIANT_GRAPH_COPIES)
        variant |= 1 << 15
    graph = {"0": 1, "1": 2}.get(env("HNET_GRAPH", ""), 0)
    return int(env("HNET_WARP_EXACT", "0") != "0"), graph, variant


FROM_FILE = -1      # include/hnet.h HNET_FROM_FILE


def make_config(variant="full", mc_samples=16, dropout_p=0.05, mc_seed=0, max_batch=1, emit_error_map=False, device_id=0, mc_shard=None, precision=None):
    """hnet_config from the arguments of HnetEngine / HnetGroup (None = HNET_FROM_FILE where the C ABI has it) and this process's kernel-selection environment"""
    cfg = Config()
    lib().hnet_default_config(C.byref(cfg))
    if precision is None:   # default: the library's (fp16 planes, fp32-grade); HNET_PRECISION=2 / 0 select split-bf16 / the exact-fp32 MFMA path
        precision = cfg.precision if IGNORE_ENV else int(os.environ.get("HNET_PRECISION", str(cfg.precision)))
    cfg.device_id = device_id
    cfg.use_prior, cfg.blocks_to_run = VARIANTS[variant] if variant is not None else (FROM_FILE, FROM_FILE)
    cfg.mc_samples = FROM_FILE if mc_samples is None else mc_samples
    cfg.dropout_p = -1.0 if dropout_p is None else dropout_p
    cfg.mc_seed = mc_seed
    cfg.emit_error_map, cfg.precision, cfg.max_batch = (FROM_FILE if emit_error_map is None else int(emit_error_map)), precision, max_batch
    if mc_shard is not None:
        cfg.mc_sample_begin, cfg.mc_sample_end = mc_shard
    cfg.warp_exact, cfg.graph, cfg.variant = kernel_selection_from_env()
    return cfg


class HnetEngine:
    """One hnet context (one GPU).  `weights` is an HNETW001 blob (bytes) or a path to one."""

    def __init__(self, weights, variant="full", mc_samples=16, dropout_p=0.05, mc_seed=0, max_batch=1,
                 emit_error_map=False, device_id=0, mc_shard=None, precision=None):
        """variant / mc_samples / dropout_p / emit_error_map = None: HNET_FROM_FILE - taken from the blob's `hnet.variant` record
        (cuahn_vio_amd.weights.pack_state_dict(..., variant=...)); `config()` returns what is in effect"""
        L = lib()
        cfg = make_config(variant, mc_samples, dropout_p, mc_seed, max_batch, emit_error_map, device_id, mc_shard, precision)
        self.cfg, self.variant, self._L = cfg, variant, L
        self._owned = True
        self._h = C.c_void_p()
        if isinstance(weights, (bytes, bytearray)):
            buf = (C.c_char * len(weights)).from_buffer_copy(weights)
            rc = L.hnet_create_from_memory(C.byref(cfg), C.cast(buf, C.c_void_p), len(weights), C.byref(self._h))
        else:
            rc = L.hnet_create(C.byref(cfg), str(weights).encode(), C.byref(self._h))
        check(None, rc)
        used = self.config()
        if variant is None:
            self.variant = {(0, 3): "full", (1, 3): "prior3", (1, 2): "prior2", (1, 1): "prior1"}[(used.use_prior, used.blocks_to_run if used.use_prior else 3)]
        self.n_local = (cfg.mc_sample_end - cfg.mc_sample_begin) if mc_shard else used.mc_samples

    def config(self):
        """hnet_get_config: the configuration in effect (HNET_FROM_FILE fields resolved from the blob)"""
        out = Config()
        check(self._h, self._L.hnet_get_config(self._h, C.byref(out)))
        return out

    def precision(self):
        """the arithmetic mode in effect (hnet_precision: HNET_PREC_F16X2 falls back to HNET_PREC_BF16X3 outside the fp16 range)"""
        return int(self._L.hnet_precision(self._h))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._L.hnet_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @classmethod
    def _borrow(cls, handle, variant=None):
        """a view of a context somebody else owns (a member of an HnetGroup): every method works, close() does not destroy"""
        e = cls.__new__(cls)
        e._L, e._h, e._owned, e.variant = lib(), C.c_void_p(handle), False, variant
        e.cfg = e.config()
        e.n_local = (e.cfg.mc_sample_end - e.cfg.mc_sample_begin) or e.cfg.mc_samples
        return e

    @property
    def handle(self):
        return self._h

    # ---- batched inference, host buffers -------------------------------------------------------
    def infer_batch(self, prev, curr, prior=None, pair_seq0=0, want_err=False):
        prev, curr = np.ascontiguousarray(prev), np.ascontiguousarray(curr)
        if prev.dtype != curr.dtype or prev.dtype not in (np.uint8, np.float32):
            raise TypeError("images must both be uint8 or float32")
        fmt = PIX_U8 if prev.dtype == np.uint8 else PIX_F32
        b = prev.reshape(-1, IMG_H, IMG_W).shape[0]
        mean = np.zeros((b, 8), np.float32)
        cov = np.zeros((b, 8, 8), np.float32)
        err = np.zeros((b, IMG_H, IMG_W), np.float32) if want_err else None
        pr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float32).reshape(b, 8)
        rc = self._L.hnet_infer_batch(self._h, prev.ctypes.data, curr.ctypes.data, fmt, _fp(pr) if pr is not None else None,
                                      b, pair_seq0, _fp(mean), _fp(cov), _fp(err) if want_err else None)
        check(self._h, rc)
        return (mean, cov, err) if want_err else (mean, cov)

    # ---- device-resident entry points (raw device pointers, e.g. torch tensors' data_ptr()) ------
    @staticmethod
    def _stream(stream):
        """stream argument of the device entry points: None -> the context's own stream (C ABI: NULL); a torch.cuda.Stream
        or a raw hipStream_t handle -> that stream.  The handle 0 is torch's name for the legacy default stream; the C ABI
        reserves NULL for "context stream", so 0 is passed as hipStreamLegacy ((hipStream_t)1) — the kernels then run in
        order with torch's default-stream work (tensor fills, RCCL collectives)."""
        if stream is None:
            return None
        h = getattr(stream, "cuda_stream", stream)
        return HIP_STREAM_LEGACY if int(h) == 0 else int(h)

    def infer_batch_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov, d_err=None, stream=None):
        check(self._h, self._L.hnet_infer_batch_device(self._h, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov,
                                                       d_err, self._stream(stream)))

    def infer_batch_packed_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_out72, d_err=None, stream=None):
        """the same forward with the packed [batch, 72] record (mean | cov) as output: the message of the multi-GPU gather, written by the ensemble kernel"""
        check(self._h, self._L.hnet_infer_batch_packed_device(self._h, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_out72, d_err, self._stream(stream)))

    def mc_finish_packed_device(self, d_mean_s, d_logvar_s, n_total, d_h1, batch, d_out72, stream=None):
        check(self._h, self._L.hnet_mc_finish_packed_device(self._h, d_mean_s, d_logvar_s, n_total, d_h1, batch, d_out72, self._stream(stream)))

    def infer_mc_partial_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean_s, d_logvar_s, d_h1, stream=None):
        check(self._h, self._L.hnet_infer_mc_partial_device(self._h, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean_s,
                                                            d_logvar_s, d_h1, self._stream(stream)))

    def mc_finish_gathered_device(self, d_gathered, world, n_local, d_h1, batch, d_out72, stream=None):
        """hnet_mc_finish_gathered_device: the ensemble straight from the all-gathered [world][2][B][n_local][8] buffer"""
        check(self._h, self._L.hnet_mc_finish_gathered_device(self._h, d_gathered, world, n_local, d_h1, batch, d_out72, self._stream(stream)))

    def mc_finish_device(self, d_mean_s, d_logvar_s, n_total, d_h1, batch, d_mean, d_cov, stream=None):
        check(self._h, self._L.hnet_mc_finish_device(self._h, d_mean_s, d_logvar_s, n_total, d_h1, batch, d_mean, d_cov,
                                                     self._stream(stream)))

    def synchronize(self, stream=None):
        check(self._h, self._L.hnet_synchronize(self._h, self._stream(stream)))

    def overflow_flag(self, stream=None):
        """hnet_overflow_flag: synchronises, returns and clears the device word the forwards OR into when an output is not finite (bit 0).
        The device-resident entry points cannot look at their results; in HNET_PREC_F16X2 a set bit with finite inputs means an
        activation left the fp16-plane range and the batch should be repeated on a HNET_PREC_BF16X3 context."""
        v = C.c_int(0)
        check(self._h, self._L.hnet_overflow_flag(self._h, self._stream(stream), C.byref(v)))
        return int(v.value)

    def time_batch_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov, iters):
        per = np.zeros(iters, np.float32)
        tot = C.c_float(0)
        check(self._h, self._L.hnet_time_batch_device(self._h, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov,
                                                      iters, _fp(per), C.byref(tot)))
        return per, float(tot.value)

    def stages(self):
        n = self._L.hnet_stage_count(self._h)
        return [(self._L.hnet_stage_name(self._h, i).decode(), self._L.hnet_stage_flops_per_pair(self._h, i)) for i in range(n)]

    def stage_kernels(self):
        """kernels per stage of the last profiled forward (hnet_stage_kernels: a split-K layer with a separate reduce launch counts 2)"""
        return [int(self._L.hnet_stage_kernels(self._h, i)) for i in range(self._L.hnet_stage_count(self._h))]

    def profile_batch_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov, iters):
        ms = np.zeros(self._L.hnet_stage_count(self._h), np.float32)
        check(self._h, self._L.hnet_profile_batch_device(self._h, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_mean, d_cov,
                                                         iters, _fp(ms)))
        return ms

    def last_timing(self):
        t = Timing()
        check(self._h, self._L.hnet_last_timing(self._h, C.byref(t)))
        return {"device_ms": t.device_ms, "host_ms": t.host_ms, "n_inferences": t.n_inferences,
                "sum_device_ms_after_100": t.sum_device_ms_after_100, "n_main_inferences": t.n_main_inferences}

    # ---- operator-level entry points (NCHW host arrays, like the reference tensors) ---------------
    def op_warp(self, img, h):
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(IMG_H, IMG_W)
        hm = np.ascontiguousarray(h, dtype=np.float32).reshape(9)
        out = np.zeros((IMG_H, IMG_W), np.float32)
        check(self._h, self._L.hnet_op_warp(self._h, _fp(img), _fp(hm), _fp(out)))
        return out

    def op_dlt(self, dst):
        d = np.ascontiguousarray(dst, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((d.shape[0], 3, 3), np.float32)
        check(self._h, self._L.hnet_op_dlt(self._h, _fp(d), d.shape[0], _fp(out)))
        return out

    def op_conv(self, layer, x):
        from .weights import CONV_LAYERS
        x = np.ascontiguousarray(x, dtype=np.float32)
        b, cin, h, w = x.shape
        _n, lcin, cout, k, s = CONV_LAYERS[layer]
        assert cin == lcin
        p = (k - 1) // 2
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out = np.zeros((b, cout, ho, wo), np.float32)
        check(self._h, self._L.hnet_op_conv(self._h, layer, _fp(x), b, h, w, _fp(out)))
        return out

    def op_block4_fused(self, x, reverse=False):
        """the fused block_4_0 + block_4_1 kernel alone: x [B,2,224,320] -> [B,16,112,160]"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        b = x.shape[0]
        assert x.shape[1:] == (2, IMG_H, IMG_W)
        out = np.zeros((b, 16, IMG_H // 2, IMG_W // 2), np.float32)
        check(self._h, self._L.hnet_op_block4_fused(self._h, _fp(x), b, int(bool(reverse)), _fp(out)))
        return out

    def op_block3_fused(self, x):
        """the fused block_3_0 + block_3_1 kernel alone (fp16-plane mode): x [B,2,112,160] -> [B,32,56,80]"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        b = x.shape[0]
        assert x.shape[1:] == (2, IMG_H // 2, IMG_W // 2)
        out = np.zeros((b, 32, IMG_H // 4, IMG_W // 4), np.float32)
        check(self._h, self._L.hnet_op_block3_fused(self._h, _fp(x), b, _fp(out)))
        return out

    def op_block42_fused(self, x):
        """the fused block_4_2 + block_4_3 kernel alone (fp16-plane mode): x [B,16,112,160] -> [B,64,28,40]"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        b = x.shape[0]
        assert x.shape[1:] == (16, IMG_H // 2, IMG_W // 2)
        out = np.zeros((b, 64, IMG_H // 8, IMG_W // 8), np.float32)
        check(self._h, self._L.hnet_op_block42_fused(self._h, _fp(x), b, _fp(out)))
        return out

    def op_prep(self, img1, img2, h, k):
        i1 = np.ascontiguousarray(img1, dtype=np.float32).reshape(IMG_H, IMG_W)
        i2 = np.ascontiguousarray(img2, dtype=np.float32).reshape(IMG_H, IMG_W)
        hm = None if h is None else np.ascontiguousarray(h, dtype=np.float32).reshape(9)
        out = np.zeros((2, IMG_H // k, IMG_W // k), np.float32)
        check(self._h, self._L.hnet_op_prep(self._h, _fp(i1), _fp(i2), _fp(hm) if hm is not None else None, k, _fp(out)))
        return out

    # ---- image pre-processing (CamBase.h:165-186)
    def set_camera(self, k, d, raw_rows, raw_cols, fisheye=True):
        cam = _capi.Camera(int(bool(fisheye)), int(raw_rows), int(raw_cols), (C.c_double * 4)(*[float(x) for x in k]),
                           (C.c_double * 4)(*[float(x) for x in d]))
        check(self._h, self._L.hnet_set_camera(self._h, C.byref(cam)))

    def set_undistort_maps(self, map_x, map_y, raw_rows, raw_cols):
        mx = np.ascontiguousarray(map_x, dtype=np.float32).reshape(IMG_H, IMG_W)
        my = np.ascontiguousarray(map_y, dtype=np.float32).reshape(IMG_H, IMG_W)
        check(self._h, self._L.hnet_set_undistort_maps(self._h, _fp(mx), _fp(my), int(raw_rows), int(raw_cols)))

    def get_undistort_maps(self):
        mx, my = np.zeros((IMG_H, IMG_W), np.float32), np.zeros((IMG_H, IMG_W), np.float32)
        check(self._h, self._L.hnet_get_undistort_maps(self._h, _fp(mx), _fp(my)))
        return mx, my

    def push_raw_image(self, raw, time_stamp):
        raw = np.asarray(raw)
        if raw.dtype != np.uint8 or raw.ndim != 2:
            raise ValueError("expected a 2-D uint8 image")
        if raw.strides[1] != 1:
            raw = np.ascontiguousarray(raw)
        check(self._h, self._L.hnet_push_raw_image(self._h, raw.ctypes.data, raw.shape[0], raw.shape[1], raw.strides[0], float(time_stamp)))

    def op_undistort(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        out = np.zeros((IMG_H, IMG_W), np.uint8)
        check(self._h, self._L.hnet_op_undistort(self._h, raw.ctypes.data, raw.shape[0], raw.shape[1], raw.strides[0], out.ctypes.data))
        return out

    def op_prep_u8(self, img1, img2, h, k):
        i1 = np.ascontiguousarray(img1, dtype=np.uint8).reshape(IMG_H, IMG_W)
        i2 = np.ascontiguousarray(img2, dtype=np.uint8).reshape(IMG_H, IMG_W)
        hm = None if h is None else np.ascontiguousarray(h, dtype=np.float32).reshape(9)
        out = np.zeros((2, IMG_H // k, IMG_W // k), np.float32)
        u8p = C.POINTER(C.c_uint8)
        check(self._h, self._L.hnet_op_prep_u8(self._h, i1.ctypes.data_as(u8p), i2.ctypes.data_as(u8p),
                                               _fp(hm) if hm is not None else None, k, _fp(out)))
        return out

    def debug_layer_output(self, layer, pair=0):
        """[Cout, Ho, Wo] output of conv layer `layer` from the last forward"""
        from .weights import CONV_LAYERS
        buf = np.zeros(8 * IMG_H * IMG_W, np.float32)
        check(self._h, self._L.hnet_debug_layer_output(self._h, layer, pair, _fp(buf), buf.size))
        h, w = {1: (28, 40), 2: (56, 80), 3: (112, 160), 4: (224, 320)}[int(CONV_LAYERS[layer][0][6])]
        for name, _cin, cout, k, s in CONV_LAYERS:
            if name[6] != CONV_LAYERS[layer][0][6]:
                continue
            p = (k - 1) // 2
            h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            if name == CONV_LAYERS[layer][0]:
                return buf[: cout * h * w].reshape(cout, h, w).copy()
        raise AssertionError

    def debug_h_part1(self, pair=0):
        out = np.zeros(9, np.float32)
        check(self._h, self._L.hnet_debug_h_part1(self._h, pair, _fp(out)))
        return out.reshape(3, 3)


class HnetGroup:
    """hnet_group (include/hnet.h): n_ctx contexts of one configuration for INDEPENDENT steps - step k runs on member k mod n_ctx, each member on its own HIP
    stream (created first, each on its own priority level / hardware queue).  `weights`: an HNETW001 blob (bytes) or a path."""

    def __init__(self, weights, n_ctx, variant="full", mc_samples=16, dropout_p=0.05, mc_seed=0, max_batch=1, emit_error_map=False, device_id=0, precision=None):
        L = lib()
        cfg = make_config(variant, mc_samples, dropout_p, mc_seed, max_batch, emit_error_map, device_id, None, precision)
        self._L, self._g, self.variant = L, C.c_void_p(), variant
        if isinstance(weights, (bytes, bytearray)):
            buf = (C.c_char * len(weights)).from_buffer_copy(weights)
            rc = L.hnet_create_group_from_memory(C.byref(cfg), C.cast(buf, C.c_void_p), len(weights), int(n_ctx), C.byref(self._g))
        else:
            rc = L.hnet_create_group(C.byref(cfg), str(weights).encode(), int(n_ctx), C.byref(self._g))
        check(None, rc)
        self.n = int(L.hnet_group_size(self._g))
        self.members = [HnetEngine._borrow(L.hnet_group_context(self._g, i), variant) for i in range(self.n)]

    def _check(self, rc):
        if rc != _capi.HNET_OK:
            raise HnetError(rc, self._L.hnet_status_string(rc).decode() + ": " + self._L.hnet_group_last_error(self._g).decode())

    def stream(self, i):
        """member i's hipStream_t as an integer handle (torch.cuda.ExternalStream(handle) wraps it)"""
        return int(self._L.hnet_group_stream(self._g, i) or 0)

    def infer_batch_packed_device(self, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_out72, d_err=None):
        """the next step, on the next member's stream; returns that member's index.  Does not synchronise; consecutive steps need distinct output buffers"""
        m = C.c_int(-1)
        self._check(self._L.hnet_group_infer_batch_packed_device(self._g, d_prev, d_curr, fmt, d_prior, batch, pair_seq0, d_out72, d_err, C.byref(m)))
        return int(m.value)

    def join(self, stream):
        """`stream` (a torch stream or a raw handle) waits for everything enqueued on the members so far"""
        self._check(self._L.hnet_group_join(self._g, HnetEngine._stream(stream)))

    def synchronize(self):
        self._check(self._L.hnet_group_synchronize(self._g))

    def overflow_flag(self):
        v = C.c_int(0)
        self._check(self._L.hnet_group_overflow_flag(self._g, C.byref(v)))
        return int(v.value)

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()
            self._L.hnet_destroy_group(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HomographyNet:
    """Mirror of ``pytorch::HomographyNet`` (HomographyNet.h:23-67).

    HomographyNet(network_model_path, network_model_iterative_path, use_prior, num_of_iteration, show_imgs)
    — the model path names an HNETW001 weight blob instead of a TorchScript file; as in the reference the
    substring "_showError" in the path selects the variant that also produces the photometric error map
    (HomographyNet.cpp:96-100).  The extra keyword arguments expose what the reference freezes at trace time.
    """

    def __init__(self, network_model_path, network_model_iterative_path="", use_prior=False, num_of_iteration=1,
                 show_imgs=False, *, blocks_to_run=None, mc_samples=None, dropout_p=None, mc_seed=0, device_id=0,
                 weights_blob=None, precision=None, blocks_to_run_iterative=None, weights_blob_iterative=None):
        """blocks_to_run / mc_samples / dropout_p = None: from the file's `hnet.variant` record (what the reference freezes into the traced .pt;
        a record-less blob gives the reference's launch values 3 / 16 / 0.05) - explicit values are overrides, as HNET_* are for the C++ adapter"""
        self.use_prior_4pt_offset = bool(use_prior)
        self.cv_imshow = bool(show_imgs)
        self.iteration = num_of_iteration > 1
        main_err, iter_err = "_showError" in str(network_model_path), "_showError" in str(network_model_iterative_path)

        def make(weights, blocks, emit):
            variant = "full" if not use_prior else (None if blocks is None else {3: "prior3", 2: "prior2", 1: "prior1"}[blocks])
            e = HnetEngine(weights, variant=variant, mc_samples=mc_samples, dropout_p=dropout_p, mc_seed=mc_seed, max_batch=1,
                           emit_error_map=emit, device_id=device_id, precision=precision)
            u = e.config()
            if bool(u.use_prior) != bool(use_prior):
                e.close()
                raise ValueError("use_prior disagrees with the variant recorded in the weight file")
            print(f"model variant: {e.variant}, MC-dropout N = {u.mc_samples}, p = {u.dropout_p:g}, error map {'on' if u.emit_error_map else 'off'}")
            return e, bool(u.emit_error_map)

        print("Loading the Network Model (HNETW001 weights) ...")
        # ("_showError" in the name forces the map like the reference's sniff; otherwise the file's record decides; unused with an iterative model, :199)
        self._eng, any_err = make(weights_blob if weights_blob is not None else network_model_path, blocks_to_run,
                                  False if self.iteration else (True if main_err else None))
        t = self._eng.last_timing()
        print(f"[TIME]: {t['host_ms']:.4f} milliseconds for the first network inference")
        self._eng_iter = None
        if self.iteration:      # HomographyNet.cpp:20-24: a second model for iteration > 0, warmed up (:49-56), fed the same frames
            w_it = weights_blob_iterative if weights_blob_iterative is not None else (weights_blob if weights_blob is not None else network_model_iterative_path)
            self._eng_iter, any_err = make(w_it, blocks_to_run_iterative, True if iter_err else None)
            check(self._eng_iter.handle, self._eng._L.hnet_attach_images(self._eng_iter.handle, self._eng.handle))
            print("IEKF! Load the Network for Iteration!")
        self.show_phtometric_error = any_err       # one member in the reference: the file loaded last decides (:96-100, :117-121)
        self._pred_mean = np.zeros((8, 1), np.float32)
        self._pred_Cov = np.zeros((8, 8), np.float32)
        self.last_error_map = None

    @property
    def img_counter(self):
        return self._eng._L.hnet_image_count(self._eng.handle)

    def load_current_img(self, img, time_stamp):
        """HomographyNet.cpp:127-151 — img: 224x320 uint8 (a cv::Mat CV_8UC1 in the reference)"""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.shape != (IMG_H, IMG_W):
            raise ValueError("expected a 224x320 uint8 image")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        if self.img_counter == 0:
            print("First Image Comes into the Network Object!")
        check(self._eng.handle, self._eng._L.hnet_push_image(self._eng.handle, img.ctypes.data, IMG_H, IMG_W,
                                                              img.strides[0], float(time_stamp)))

    def network_inference(self, prior_4pt_offset_vec, num_of_inference=0):
        """HomographyNet.cpp:153-252"""
        mean = np.zeros(8, np.float32)
        cov = np.zeros((8, 8), np.float32)
        net = self._eng_iter if (self._eng_iter is not None and num_of_inference > 0) else self._eng      # HomographyNet_model / _model_iterative (:183, :211)
        err = np.zeros((IMG_H, IMG_W), np.uint8) if (self.show_phtometric_error and (net is self._eng_iter or not self.iteration)) else None
        pr = None
        if self.use_prior_4pt_offset:
            pr = np.ascontiguousarray(prior_4pt_offset_vec, dtype=np.float64).reshape(8)
        rc = self._eng._L.hnet_infer(net.handle, pr.ctypes.data_as(C.POINTER(C.c_double)) if pr is not None else None,
                                     int(num_of_inference), _fp(mean), _fp(cov),
                                     err.ctypes.data_as(C.POINTER(C.c_uint8)) if err is not None else None)
        if rc == _capi.ERR_NOT_READY:   # :155-158 prints and returns with the previous outputs
            print("HNet cannot inference! Only has one image!")
            return
        check(net.handle, rc)
        self._pred_mean = mean.reshape(8, 1)
        self._pred_Cov = cov
        self.last_error_map = err
        if num_of_inference == 0:
            t = self._eng.last_timing()
            if t["n_main_inferences"] > 100:
                avg = t["sum_device_ms_after_100"] / (t["n_main_inferences"] - 100)
                print(f"[TIME]: {t['device_ms']:.3f} (avg. = {avg:.3f}) milliseconds for pure network inference")

    def get_pred_mean(self):
        return self._pred_mean.astype(np.float64)

    def get_pred_Cov(self):
        return self._pred_Cov.astype(np.float64)

    def get_latest_inference_time(self):
        return self._eng._L.hnet_latest_time(self._eng.handle)

    def close(self):
        """the iterative model's context reads the main context's frames (hnet_attach_images): it goes first"""
        if getattr(self, "_eng_iter", None) is not None:
            self._eng_iter.close()
            self._eng_iter = None
        if getattr(self, "_eng", None) is not None:
            self._eng.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
