"""ctypes binding of libhnet_hip.so — the C ABI declared in include/hnet.h.

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# HNET_LIB_PATH: another build of the SAME library (same-box A/B of two source states with tools/ab_bench.py); never a fallback
LIB_PATH = os.environ.get("HNET_LIB_PATH") or os.path.join(_PKG, "libhnet_hip.so")

HNET_OK = 0
PREC_FP32, PREC_BF16, PREC_BF16X3, PREC_F16X2 = 0, 1, 2, 3
PIX_U8, PIX_F32 = 0, 1
ERR_NOT_READY = 4

# every symbol include/hnet.h declares (tests check the library exports all of them)
SYMBOLS = [
    "hnet_default_config", "hnet_create", "hnet_create_from_memory", "hnet_destroy", "hnet_status_string",
    "hnet_last_error", "hnet_version", "hnet_push_image", "hnet_attach_images", "hnet_image_count", "hnet_latest_time", "hnet_infer",
    "hnet_infer_batch", "hnet_infer_batch_device", "hnet_infer_batch_packed_device", "hnet_infer_mc_partial_device", "hnet_mc_finish_device",
    "hnet_mc_finish_packed_device", "hnet_mc_finish_gathered_device",
    "hnet_synchronize", "hnet_last_timing", "hnet_time_batch_device", "hnet_stage_count", "hnet_stage_name",
    "hnet_stage_flops_per_pair", "hnet_stage_kernels", "hnet_get_config", "hnet_profile_batch_device", "hnet_op_warp", "hnet_op_dlt", "hnet_op_conv",
    "hnet_op_prep", "hnet_op_prep_u8", "hnet_debug_layer_output", "hnet_debug_h_part1",
    "hnet_set_camera", "hnet_set_undistort_maps", "hnet_get_undistort_maps", "hnet_push_raw_image", "hnet_op_undistort",
    "hnet_op_block4_fused", "hnet_op_block3_fused", "hnet_op_block42_fused", "hnet_precision", "hnet_overflow_flag",
    "hnet_create_group", "hnet_create_group_from_memory", "hnet_destroy_group", "hnet_group_size", "hnet_group_context", "hnet_group_stream",
    "hnet_group_last_error", "hnet_group_infer_batch_packed_device", "hnet_group_join", "hnet_group_synchronize", "hnet_group_overflow_flag",
]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device_id", C.c_int32), ("use_prior", C.c_int32),
                ("blocks_to_run", C.c_int32), ("mc_samples", C.c_int32), ("dropout_p", C.c_float),
                ("mc_seed", C.c_uint64), ("emit_error_map", C.c_int32), ("precision", C.c_int32),
                ("max_batch", C.c_int32), ("mc_sample_begin", C.c_int32), ("mc_sample_end", C.c_int32),
                ("warp_exact", C.c_int32), ("graph", C.c_int32), ("variant", C.c_uint32)]


class Camera(C.Structure):
    """hnet_camera: fisheye flag, raw size, (fx, fy, cx, cy), distortion (k1..k4 or k1, k2, p1, p2)"""
    _fields_ = [("fisheye", C.c_int32), ("raw_rows", C.c_int32), ("raw_cols", C.c_int32), ("k", C.c_double * 4), ("d", C.c_double * 4)]


class Timing(C.Structure):
    _fields_ = [("device_ms", C.c_double), ("host_ms", C.c_double), ("n_inferences", C.c_int64),
                ("sum_device_ms_after_100", C.c_double), ("n_main_inferences", C.c_int64)]


class HnetError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"hnet status {status}: {msg}")
        self.status = status


_lib = None


def lib():
    """load libhnet_hip.so (built by __graft_entry__.build() / csrc/Makefile); fail loudly when absent"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, fp, u8p, dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    L.hnet_default_config.argtypes = [C.POINTER(Config)]
    L.hnet_default_config.restype = None
    L.hnet_create.argtypes = [C.POINTER(Config), C.c_char_p, C.POINTER(vp)]
    L.hnet_create_from_memory.argtypes = [C.POINTER(Config), vp, C.c_size_t, C.POINTER(vp)]
    L.hnet_destroy.argtypes = [vp]
    L.hnet_destroy.restype = None
    L.hnet_status_string.argtypes = [C.c_int]
    L.hnet_status_string.restype = C.c_char_p
    L.hnet_last_error.argtypes = [vp]
    L.hnet_last_error.restype = C.c_char_p
    L.hnet_version.restype = C.c_char_p
    L.hnet_push_image.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    L.hnet_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.hnet_set_undistort_maps.argtypes = [vp, fp, fp, C.c_int, C.c_int]
    L.hnet_get_undistort_maps.argtypes = [vp, fp, fp]
    L.hnet_push_raw_image.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    L.hnet_op_undistort.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.hnet_attach_images.argtypes = [vp, vp]
    L.hnet_image_count.argtypes = [vp]
    L.hnet_precision.argtypes = [vp]
    L.hnet_latest_time.argtypes = [vp]
    L.hnet_latest_time.restype = C.c_double
    L.hnet_infer.argtypes = [vp, dp, C.c_int, fp, fp, u8p]
    L.hnet_infer_batch.argtypes = [vp, vp, vp, C.c_int, fp, C.c_int, C.c_uint64, fp, fp, fp]
    L.hnet_infer_batch_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, vp, vp]
    L.hnet_infer_mc_partial_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, vp, vp]
    L.hnet_mc_finish_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.hnet_infer_batch_packed_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, vp]
    L.hnet_mc_finish_packed_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
    L.hnet_mc_finish_gathered_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    L.hnet_get_config.argtypes = [vp, C.POINTER(Config)]
    L.hnet_synchronize.argtypes = [vp, vp]
    L.hnet_overflow_flag.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.hnet_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.hnet_time_batch_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, C.c_int, fp, fp]
    L.hnet_stage_count.argtypes = [vp]
    L.hnet_stage_name.argtypes = [vp, C.c_int]
    L.hnet_stage_name.restype = C.c_char_p
    L.hnet_stage_flops_per_pair.argtypes = [vp, C.c_int]
    L.hnet_stage_kernels.argtypes = [vp, C.c_int]
    L.hnet_stage_flops_per_pair.restype = C.c_double
    L.hnet_profile_batch_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, C.c_int, fp]
    L.hnet_op_warp.argtypes = [vp, fp, fp, fp]
    L.hnet_op_dlt.argtypes = [vp, fp, C.c_int, fp]
    L.hnet_op_conv.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp]
    L.hnet_op_block4_fused.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.hnet_op_block3_fused.argtypes = [vp, fp, C.c_int, fp]
    L.hnet_op_block42_fused.argtypes = [vp, fp, C.c_int, fp]
    L.hnet_op_prep.argtypes = [vp, fp, fp, fp, C.c_int, fp]
    L.hnet_op_prep_u8.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), fp, C.c_int, fp]
    L.hnet_debug_layer_output.argtypes = [vp, C.c_int, C.c_int, fp, C.c_size_t]
    L.hnet_debug_h_part1.argtypes = [vp, C.c_int, fp]
    L.hnet_create_group.argtypes = [C.POINTER(Config), C.c_char_p, C.c_int, C.POINTER(vp)]
    L.hnet_create_group_from_memory.argtypes = [C.POINTER(Config), vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.hnet_destroy_group.argtypes = [vp]
    L.hnet_destroy_group.restype = None
    L.hnet_group_size.argtypes = [vp]
    L.hnet_group_context.argtypes = [vp, C.c_int]
    L.hnet_group_context.restype = vp
    L.hnet_group_stream.argtypes = [vp, C.c_int]
    L.hnet_group_stream.restype = vp
    L.hnet_group_last_error.argtypes = [vp]
    L.hnet_group_last_error.restype = C.c_char_p
    L.hnet_group_infer_batch_packed_device.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_uint64, vp, vp, C.POINTER(C.c_int)]
    L.hnet_group_join.argtypes = [vp, vp]
    L.hnet_group_synchronize.argtypes = [vp]
    L.hnet_group_overflow_flag.argtypes = [vp, C.POINTER(C.c_int)]
    for name in SYMBOLS:
        getattr(L, name)   # AttributeError here = the library does not export what include/hnet.h declares
    _lib = L
    return L


def check(ctx, status):
    if status != HNET_OK:
        L = lib()
        msg = L.hnet_status_string(status).decode()
        detail = L.hnet_last_error(ctx).decode() if ctx else ""
        raise HnetError(status, f"{msg}{': ' + detail if detail else ''}")
