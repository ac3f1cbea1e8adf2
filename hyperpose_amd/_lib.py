"""ctypes binding of libhp_hip.so (include/hp_hip.h).  Fails loudly: there is NO CPU fallback.

The shared library is built in-tree by ``python -m hyperpose_amd.build`` (hipcc, gfx950).  Importing this
module never needs a GPU; calling into it does (hp_init reports HP_ERR_NO_DEVICE otherwise).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhp_hip.so")

HP_OK = 0
HP_ERR_INVALID, HP_ERR_HIP, HP_ERR_CAPACITY, HP_ERR_STATE, HP_ERR_NO_DEVICE = -1, -2, -3, -4, -5


class HpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libhp_hip error {code}: {msg}")
        self.code = code


class BodyPart(C.Structure):
    _fields_ = [("has_value", C.c_int32), ("x", C.c_float), ("y", C.c_float), ("score", C.c_float)]


class Human(C.Structure):
    _fields_ = [("parts", BodyPart * 18), ("score", C.c_float)]


class Peak(C.Structure):
    _fields_ = [("part_id", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("score", C.c_float), ("id", C.c_int32)]


class Conn(C.Structure):
    _fields_ = [("pair_id", C.c_int32), ("cid1", C.c_int32), ("cid2", C.c_int32), ("score", C.c_float)]


PART_DTYPE = np.dtype([("has_value", "<i4"), ("x", "<f4"), ("y", "<f4"), ("score", "<f4")])
HUMAN_DTYPE = np.dtype({"names": ["parts", "score"], "formats": [(PART_DTYPE, 18), "<f4"]})
PEAK_DTYPE = np.dtype([("part_id", "<i4"), ("x", "<i4"), ("y", "<i4"), ("score", "<f4"), ("id", "<i4")])
CONN_DTYPE = np.dtype([("pair_id", "<i4"), ("cid1", "<i4"), ("cid2", "<i4"), ("score", "<f4")])
assert HUMAN_DTYPE.itemsize == C.sizeof(Human) == 292

_lib = None

# every symbol include/hp_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "hp_init", "hp_device_count", "hp_last_error", "hp_version", "hp_malloc", "hp_free", "hp_malloc_host",
    "hp_free_host", "hp_memcpy_h2d", "hp_memcpy_d2h", "hp_device_synchronize", "hp_stream_wait_stream", "hp_dist_unique_id", "hp_dist_init", "hp_dist_destroy", "hp_dist_broadcast_weights", "hp_dist_shard", "hp_preproc_u8hwc_to_f32nchw", "hp_resize_u8c3", "hp_letterbox_u8c3", "hp_letterbox_inner", "hp_resume_ratio",
    "hp_paf_create", "hp_paf_stream", "hp_paf_destroy", "hp_paf_set_conf_thresh", "hp_paf_set_paf_thresh", "hp_paf_process_batch",
    "hp_paf_enqueue", "hp_paf_collect", "hp_paf_debug_peaks", "hp_paf_debug_conns", "hp_paf_debug_maps", "hp_paf_debug_sort",
    "hp_pifpaf_create", "hp_pifpaf_destroy", "hp_pifpaf_process_batch", "hp_pifpaf_stream", "hp_pifpaf_enqueue", "hp_pifpaf_collect", "hp_pifpaf_decode_flags",
    "hp_ppn_create", "hp_ppn_destroy", "hp_ppn_set_thresholds", "hp_ppn_process_batch", "hp_ppn_stream", "hp_ppn_enqueue", "hp_ppn_collect", "hp_ppn_decode_flags",
    "hp_engine_create", "hp_engine_destroy", "hp_engine_max_batch", "hp_engine_describe", "hp_engine_input_size", "hp_engine_infer_u8",
    "hp_engine_infer_f32", "hp_engine_synchronize", "hp_engine_stream", "hp_engine_set_graph", "hp_engine_set_concurrency", "hp_engine_arena_info", "hp_debug_first_conv_verify", "hp_engine_concurrency", "hp_engine_num_outputs",
    "hp_engine_output", "hp_engine_output_to_host", "hp_engine_debug_tensor", "hp_engine_profile", "hp_engine_profile_sequence", "hp_engine_profile_pair", "hp_model_build", "hp_model_from_onnx", "hp_model_from_onnx_file", "hp_model_weights",
    "hp_model_input_size",
    "hp_model_destroy", "hp_model_archs", "hp_model_layers", "hp_model_outputs", "hp_model_num_weights",
    "hp_model_preproc", "hp_model_flops_per_frame", "hp_model_init_weights", "hp_engine_create_from_model", "hp_engine_create_from_model_dtype", "hp_engine_dtype", "hp_engine_split_fallbacks", "hp_engine_device_bytes", "hp_engine_save", "hp_engine_load", "hp_pipeline_create", "hp_pipeline_create_ex", "hp_pipeline_destroy", "hp_pipeline_submit", "hp_pipeline_collect", "hp_pipeline_in_flight",
]


def lib() -> C.CDLL:
    """Load libhp_hip.so or raise: the product path has no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m hyperpose_amd.build` (hipcc, gfx950). "
                              "hyperpose_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.hp_last_error.restype = C.c_char_p
        L.hp_version.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise HpError(rc, lib().hp_last_error().decode("utf-8", "replace"))
    return rc


_initialised = set()


def init(device: int = 0) -> None:
    check(lib().hp_init(int(device)))
    _initialised.add(device)


class DevBuf:
    """A device allocation owned through the C ABI (no torch needed)."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().hp_malloc(C.byref(self.ptr), C.c_size_t(self.nbytes)))

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(lib().hp_memcpy_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))
        return b

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().hp_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib().hp_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def as_ptr(x):
    """Device pointer of a DevBuf, a torch CUDA tensor or a raw int."""
    if isinstance(x, DevBuf):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))
