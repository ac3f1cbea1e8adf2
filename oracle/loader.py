"""oracle/loader.py — TEST INFRASTRUCTURE ONLY.

ctypes bindings for the CPU oracles (oracle/_build/liboracle*.so = our restatements,
oracle/_ref/libhp_ref*.so = the reference's own sources compiled where they lie).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
hyperpose_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OBodyPart(C.Structure):
    _fields_ = [("has_value", C.c_int32), ("x", C.c_float), ("y", C.c_float), ("score", C.c_float)]


class OHuman(C.Structure):
    _fields_ = [("parts", OBodyPart * 18), ("score", C.c_float)]


class OPeak(C.Structure):
    _fields_ = [("part_id", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("score", C.c_float),
                ("id", C.c_int32)]


class OConn(C.Structure):
    _fields_ = [("pair_id", C.c_int32), ("cid1", C.c_int32), ("cid2", C.c_int32), ("score", C.c_float)]


HUMAN_DTYPE = np.dtype({"names": ["parts", "score"],
                        "formats": [(np.dtype([("has_value", "<i4"), ("x", "<f4"), ("y", "<f4"), ("score", "<f4")]), 18),
                                    "<f4"]})
assert HUMAN_DTYPE.itemsize == C.sizeof(OHuman) == 292

_FP = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(_FP)


def build(ref: bool = True) -> None:
    """Compile the restatements (always) and oracle/_ref (only where /root/reference is mounted)."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir(os.environ.get("HP_REFERENCE", "/root/reference")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_libs = {}


def lib(fast: bool = False, variant: int = 0):
    """variant 1 / 2: the two OpenCV restatements rounded as an FMA build may round them (paf_oracle.cpp, HP_ORACLE_VARIANT)."""
    name = f"liboracle_v{variant}.so" if variant else ("liboracle_fast.so" if fast else "liboracle.so")
    if name not in _libs:
        path = os.path.join(HERE, "_build", name)
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.oracle_resize_area.restype = C.c_int
        L.oracle_paf_process.restype = C.c_int
        _libs[name] = L
    return _libs[name]


def ref_lib(fast: bool = False):
    """The reference's own code (oracle/_ref); None when it was never built (no /root/reference)."""
    name = "libhp_ref_fast.so" if fast else "libhp_ref.so"
    if name not in _libs:
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            try:
                build(ref=True)
            except subprocess.CalledProcessError:
                pass
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_ppn_process.restype = C.c_int
        L.ref_pifpaf_process.restype = C.c_int
        if hasattr(L, "ref_paf_process"):  # a prebuilt oracle/_ref older than round 3 has no PAF entry points
            L.ref_paf_process.restype = C.c_int
            L.ref_paf_debug.restype = C.c_int
        _libs[name] = L
    return _libs[name]


# ---------------------------------------------------------------- PAF (restatement)
def resize_area(src: np.ndarray, dh: int, dw: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    c, sh, sw = src.shape
    dst = np.zeros((c, dh, dw), np.float32)
    lib().oracle_resize_area(_fp(src), c, sh, sw, _fp(dst), dh, dw)
    return dst


def smooth(src: np.ndarray, ksize: int = 17) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    c, h, w = src.shape
    dst = np.zeros_like(src)
    lib().oracle_smooth(_fp(src), c, h, w, ksize, _fp(dst))
    return dst


def max_pool_3x3(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    c, h, w = src.shape
    dst = np.zeros_like(src)
    lib().oracle_max_pool_3x3(_fp(src), c, h, w, _fp(dst))
    return dst


def gaussian_kernel(ksize: int = 17, sigma: float = 3.0) -> np.ndarray:
    out = np.zeros(ksize, np.float32)
    lib().oracle_gaussian_kernel(ksize, C.c_double(sigma), _fp(out))
    return out


def paf_process(conf: np.ndarray, paf: np.ndarray, conf_thresh: float = 0.05, paf_thresh: float = 0.05,
                res_w: int = -1, res_h: int = -1, cap_humans: int = 128, cap_peaks: int = 8192,
                cap_conns: int = 8192, fast: bool = False, variant: int = 0):
    """One frame through the restated parser::paf::process; returns (humans, peaks, conns) numpy records."""
    conf = np.ascontiguousarray(conf, np.float32)
    paf = np.ascontiguousarray(paf, np.float32)
    j, rows, cols = conf.shape
    humans = (OHuman * cap_humans)()
    peaks = (OPeak * cap_peaks)()
    conns = (OConn * cap_conns)()
    n_peaks, n_conns = C.c_int(0), C.c_int(0)
    n = lib(fast, variant).oracle_paf_process(_fp(conf), j, rows, cols, _fp(paf), paf.shape[0],
                                     C.c_float(conf_thresh), C.c_float(paf_thresh), res_w, res_h,
                                     humans, cap_humans, peaks, cap_peaks, C.byref(n_peaks),
                                     conns, cap_conns, C.byref(n_conns))
    if n < 0 or n > cap_humans or n_peaks.value > cap_peaks or n_conns.value > cap_conns:
        raise RuntimeError(f"oracle_paf_process overflow/err: humans={n} peaks={n_peaks.value} conns={n_conns.value}")
    h = np.frombuffer(humans, dtype=HUMAN_DTYPE, count=n).copy()
    p = np.frombuffer(peaks, dtype=np.dtype([("part_id", "<i4"), ("x", "<i4"), ("y", "<i4"), ("score", "<f4"),
                                             ("id", "<i4")]), count=n_peaks.value).copy()
    c = np.frombuffer(conns, dtype=np.dtype([("pair_id", "<i4"), ("cid1", "<i4"), ("cid2", "<i4"),
                                             ("score", "<f4")]), count=n_conns.value).copy()
    return h, p, c


def std_sort_greater(scores: np.ndarray) -> np.ndarray:
    """libstdc++'s std::sort with std::greater<connection_candidate> (src/paf.cpp:249) on scores in generation order."""
    scores = np.ascontiguousarray(scores, np.float32)
    order = np.zeros(len(scores), np.int32)
    lib().oracle_std_sort_greater(_fp(scores), len(scores), order.ctypes.data_as(C.POINTER(C.c_int)))
    return order


def sort_killer(n: int) -> np.ndarray:
    """A score sequence that drives this libstdc++'s introsort into its heap-sort fall-back (McIlroy's adversary)."""
    out = np.zeros(n, np.float32)
    lib().oracle_sort_killer(n, _fp(out))
    return out


def nhwc_u8_to_nchw_f32(images: np.ndarray, factor: float = 1.0 / 255, flip_rb: bool = True) -> np.ndarray:
    images = np.ascontiguousarray(images, np.uint8)
    n, h, w, _ = images.shape
    out = np.zeros((n, 3, h, w), np.float32)
    lib().oracle_nhwc_u8_to_nchw_f32(images.ctypes.data_as(C.POINTER(C.c_uint8)), n, h, w, C.c_double(factor),
                                     int(flip_rb), _fp(out))
    return out


# ---------------------------------------------------------------- reference-compiled parsers
def ref_ppn_process(tensors, net_w=384, net_h=384, point_thresh=0.10, limb_thresh=0.05, nms_thresh=0.3,
                    cap=256, fast=False):
    L = ref_lib(fast)
    if L is None:
        raise RuntimeError("oracle/_ref not built")
    t = [np.ascontiguousarray(a, np.float32) for a in tensors]
    k, gh, gw = t[0].shape
    e, nh, nw = t[6].shape[:3]
    out = (OHuman * cap)()
    n = L.ref_ppn_process(net_w, net_h, C.c_float(point_thresh), C.c_float(limb_thresh), C.c_float(nms_thresh),
                          *[_fp(a) for a in t], k, gh, gw, e, nh, nw, out, cap)
    assert 0 <= n <= cap
    return np.frombuffer(out, dtype=HUMAN_DTYPE, count=n).copy()


_PEAK_DTYPE = np.dtype([("part_id", "<i4"), ("x", "<i4"), ("y", "<i4"), ("score", "<f4"), ("id", "<i4")])
_CONN_DTYPE = np.dtype([("pair_id", "<i4"), ("cid1", "<i4"), ("cid2", "<i4"), ("score", "<f4")])


def ref_paf_process(conf, paf, conf_thresh=0.05, paf_thresh=0.05, res_w=-1, res_h=-1, cap_humans=128,
                    cap_peaks=8192, cap_conns=8192, fast=False, debug=True):
    """One frame through the REFERENCE's own parser::paf::process (src/paf.cpp compiled in oracle/_ref; only its
    two OpenCV calls are restated).  Returns (humans, peaks, conns) like paf_process(); peaks / conns come from
    the reference's own functions called stage by stage (ref_paf_debug) and are None with debug=False."""
    L = ref_lib(fast)
    if L is None or not hasattr(L, "ref_paf_process"):
        raise RuntimeError("oracle/_ref not built (or older than the PAF entry points)")
    conf = np.ascontiguousarray(conf, np.float32)
    paf = np.ascontiguousarray(paf, np.float32)
    j, rows, cols = conf.shape
    humans = (OHuman * cap_humans)()
    n = L.ref_paf_process(_fp(conf), j, rows, cols, _fp(paf), paf.shape[0], C.c_float(conf_thresh),
                          C.c_float(paf_thresh), res_w, res_h, humans, cap_humans)
    if n < 0 or n > cap_humans:
        raise RuntimeError(f"ref_paf_process overflow/err: humans={n}")
    h = np.frombuffer(humans, dtype=HUMAN_DTYPE, count=n).copy()
    if not debug:
        return h, None, None
    peaks = (OPeak * cap_peaks)()
    conns = (OConn * cap_conns)()
    n_peaks, n_conns = C.c_int(0), C.c_int(0)
    n2 = L.ref_paf_debug(_fp(conf), j, rows, cols, _fp(paf), paf.shape[0], C.c_float(conf_thresh),
                         C.c_float(paf_thresh), res_w, res_h, peaks, cap_peaks, C.byref(n_peaks), conns, cap_conns,
                         C.byref(n_conns))
    if n2 != n or n_peaks.value > cap_peaks or n_conns.value > cap_conns:
        raise RuntimeError(f"ref_paf_debug: humans={n2} vs {n}, peaks={n_peaks.value} conns={n_conns.value}")
    p = np.frombuffer(peaks, dtype=_PEAK_DTYPE, count=n_peaks.value).copy()
    c = np.frombuffer(conns, dtype=_CONN_DTYPE, count=n_conns.value).copy()
    return h, p, c


def have_ref_paf(fast: bool = False) -> bool:
    L = ref_lib(fast)
    return L is not None and hasattr(L, "ref_paf_process")


def ref_pifpaf_process(paf, pif, net_h=385, net_w=385, thresh=0.1, cap=256, fast=False):
    L = ref_lib(fast)
    if L is None:
        raise RuntimeError("oracle/_ref not built")
    paf = np.ascontiguousarray(paf, np.float32)
    pif = np.ascontiguousarray(pif, np.float32)
    fh, fw = pif.shape[-2:]
    out = (OHuman * cap)()
    n = L.ref_pifpaf_process(net_h, net_w, C.c_float(thresh), _fp(paf), _fp(pif), fh, fw, out, cap)
    assert 0 <= n <= cap
    return np.frombuffer(out, dtype=HUMAN_DTYPE, count=n).copy()


# ---------------------------------------------------------------- stream front-end (cv::resize / non_scaling_resize)
def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize(src, (dw, dh)) for an [h, w, 3] uint8 image (default INTER_LINEAR), restated."""
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw, _ = src.shape
    dst = np.zeros((dh, dw, 3), np.uint8)
    lib().oracle_resize_linear_u8c3(_u8p(src), sw, sh, sw * 3, _u8p(dst), dw, dh, dw * 3)
    return dst


def letterbox_u8(src: np.ndarray, dw: int, dh: int, bgcolor=(0, 0, 0)) -> np.ndarray:
    """hyperpose::non_scaling_resize (src/data.cpp:53-69)."""
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw, _ = src.shape
    dst = np.zeros((dh, dw, 3), np.uint8)
    lib().oracle_letterbox_u8c3(_u8p(src), sw, sh, _u8p(dst), dw, dh, int(bgcolor[0]), int(bgcolor[1]), int(bgcolor[2]))
    return dst


def letterbox_inner(sw: int, sh: int, dw: int, dh: int):
    iw, ih = C.c_int(), C.c_int()
    lib().oracle_letterbox_inner(sw, sh, dw, dh, C.byref(iw), C.byref(ih))
    return iw.value, ih.value


def resume_ratio(xy: np.ndarray, src_wh, dst_wh) -> np.ndarray:
    """resume_ratio (human.hpp:44-58) on an [n, 2] float32 array of (x, y); returns a new array."""
    out = np.ascontiguousarray(xy, np.float32).copy()
    lib().oracle_resume_ratio(_fp(out), out.shape[0], int(src_wh[0]), int(src_wh[1]), int(dst_wh[0]), int(dst_wh[1]))
    return out
