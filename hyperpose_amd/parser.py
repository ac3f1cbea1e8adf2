"""Host-side mirror of ``hyperpose::parser`` (reference include/hyperpose/operator/parser/*.hpp) over the C ABI.

``Paf`` has the constructor arguments, setters and ``process`` meaning of ``hyperpose::parser::paf``
(paf.hpp:17-93); the batched / asynchronous forms are the MI355X additions.  All arithmetic happens in
libhp_hip.so on the GPU — this file only marshals pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CONN_DTYPE, HUMAN_DTYPE, PEAK_DTYPE, Conn, Human, Peak, as_ptr, check, lib

_FP = C.POINTER(C.c_float)


class Paf:
    """``hyperpose::parser::paf`` (conf_thresh=0.05, paf_thresh=0.05, resolution_size=(-1,-1))."""

    def __init__(self, conf_thresh: float = 0.05, paf_thresh: float = 0.05, resolution_size=(-1, -1),
                 max_batch: int = 8, cap_per_frame: int = 128):
        self._h = C.c_void_p()
        self.max_batch = int(max_batch)
        self.cap = int(cap_per_frame)
        w, h = resolution_size  # cv::Size(width, height)
        check(lib().hp_paf_create(C.byref(self._h), C.c_float(conf_thresh), C.c_float(paf_thresh), int(w), int(h),
                                  self.max_batch))
        self._out = (Human * (self.max_batch * self.cap))()
        self._n = (C.c_int * self.max_batch)()
        self._pending = 0
        lib().hp_paf_stream.restype = C.c_void_p
        self.stream = lib().hp_paf_stream(self._h)

    def after(self, stream) -> None:
        """Work enqueued on the parser's own stream from now on waits for everything already enqueued on ``stream``."""
        check(lib().hp_stream_wait_stream(C.c_void_p(self.stream), C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib().hp_paf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_paf_thresh(self, thresh: float):
        check(lib().hp_paf_set_paf_thresh(self._h, C.c_float(thresh)))

    def set_conf_thresh(self, thresh: float):
        check(lib().hp_paf_set_conf_thresh(self._h, C.c_float(thresh)))

    def _humans(self, n: int):
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[f * self.cap: f * self.cap + self._n[f]].copy() for f in range(n)]

    def process(self, conf: np.ndarray, paf: np.ndarray):
        """One frame, host arrays conf [J,rows,cols], paf [2L,rows,cols] -> structured array of humans."""
        return self.process_batch(conf[None], paf[None])[0]

    def process_batch(self, conf, paf):
        """n frames at once from host numpy arrays ([n,J,rows,cols] / [n,2L,rows,cols])."""
        conf = np.ascontiguousarray(conf, np.float32)
        paf = np.ascontiguousarray(paf, np.float32)
        n = conf.shape[0]
        cs = (C.c_int * 3)(*conf.shape[1:])
        ps = (C.c_int * 3)(*paf.shape[1:])
        check(lib().hp_paf_process_batch(self._h, n, conf.ctypes.data_as(_FP), cs, paf.ctypes.data_as(_FP), ps, 0,
                                         self._out, self.cap, self._n))
        return self._humans(n)

    def process_batch_device(self, conf_dev, paf_dev, n: int, conf_shape, paf_shape):
        cs = (C.c_int * 3)(*conf_shape)
        ps = (C.c_int * 3)(*paf_shape)
        check(lib().hp_paf_process_batch(self._h, n, as_ptr(conf_dev), cs, as_ptr(paf_dev), ps, 1, self._out, self.cap,
                                         self._n))
        return self._humans(n)

    def enqueue(self, conf_dev, paf_dev, n: int, conf_shape, paf_shape, stream=None):
        cs = (C.c_int * 3)(*conf_shape)
        ps = (C.c_int * 3)(*paf_shape)
        check(lib().hp_paf_enqueue(self._h, n, as_ptr(conf_dev), cs, as_ptr(paf_dev), ps,
                                   C.c_void_p(stream) if stream else None))
        self._pending = n

    def collect(self):
        n = self._pending
        check(lib().hp_paf_collect(self._h, self._out, self.cap, self._n))
        self._pending = 0
        return self._humans(n)

    # ---- stage-wise parity taps -------------------------------------------------------------------
    def debug_peaks(self, frame: int = 0, cap: int = 16384) -> np.ndarray:
        buf = (Peak * cap)()
        n = C.c_int(0)
        check(lib().hp_paf_debug_peaks(self._h, frame, buf, cap, C.byref(n)))
        return np.frombuffer(buf, dtype=PEAK_DTYPE, count=min(n.value, cap)).copy()

    def debug_conns(self, frame: int = 0, cap: int = 16384) -> np.ndarray:
        buf = (Conn * cap)()
        n = C.c_int(0)
        check(lib().hp_paf_debug_conns(self._h, frame, buf, cap, C.byref(n)))
        return np.frombuffer(buf, dtype=CONN_DTYPE, count=min(n.value, cap)).copy()

    def debug_maps(self, conf: np.ndarray, res_h: int, res_w: int):
        conf = np.ascontiguousarray(conf, np.float32)
        cs = (C.c_int * 3)(*conf.shape)
        up = np.zeros((conf.shape[0], res_h, res_w), np.float32)
        sm = np.zeros_like(up)
        check(lib().hp_paf_debug_maps(self._h, conf.ctypes.data_as(_FP), cs, up.ctypes.data_as(_FP), sm.ctypes.data_as(_FP)))
        return up, sm


def paf_debug_sort(scores: np.ndarray):
    """The device's restated ``std::sort(..., std::greater)`` (src/paf.cpp:249) on a score sequence; returns (order, used_heap)."""
    scores = np.ascontiguousarray(scores, np.float32)
    order = np.zeros(len(scores), np.int32)
    heap = C.c_int(0)
    check(lib().hp_paf_debug_sort(scores.ctypes.data_as(_FP), len(scores), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(heap)))
    return order, bool(heap.value)


def preproc_u8hwc_to_f32nchw(images: np.ndarray, factor: float = 1.0 / 255, flip_rb: bool = True) -> np.ndarray:
    """``hyperpose::nhwc_images_append_nchw_batch`` (src/data.cpp:21-51) on the GPU, host arrays in/out."""
    images = np.ascontiguousarray(images, np.uint8)
    n, h, w, c = images.shape
    assert c == 3
    din = _lib.DevBuf.from_numpy(images)
    dout = _lib.DevBuf(n * 3 * h * w * 4)
    check(lib().hp_preproc_u8hwc_to_f32nchw(din.ptr, n, h, w, C.c_double(factor), int(flip_rb), dout.ptr, None))
    check(lib().hp_device_synchronize())
    return dout.to_numpy(np.float32, (n, 3, h, w))


class PoseProposal:
    """``hyperpose::parser::pose_proposal`` (net_resolution, point_thresh=0.10, limb_thresh=0.05, mns_thresh=0.3),
    reference include/hyperpose/operator/parser/proposal_network.hpp:17-81."""

    def __init__(self, net_resolution, point_thresh: float = 0.10, limb_thresh: float = 0.05, mns_thresh: float = 0.3,
                 max_batch: int = 32, cap_per_frame: int = 128):
        self._h = C.c_void_p()
        self.max_batch, self.cap = int(max_batch), int(cap_per_frame)
        w, h = net_resolution  # cv::Size(width, height)
        check(lib().hp_ppn_create(C.byref(self._h), int(w), int(h), C.c_float(point_thresh), C.c_float(limb_thresh),
                                  C.c_float(mns_thresh), self.max_batch))
        self._out = (Human * (self.max_batch * self.cap))()
        self._n = (C.c_int * self.max_batch)()

    def close(self):
        if self._h:
            lib().hp_ppn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_thresholds(self, point_thresh, limb_thresh, nms_thresh):
        check(lib().hp_ppn_set_thresholds(self._h, C.c_float(point_thresh), C.c_float(limb_thresh), C.c_float(nms_thresh)))

    def process_batch(self, tensors, on_device: bool = False, n: int = None, conf_shape=None, edge_shape=None):
        """tensors = [conf_point, conf_iou, x, y, w, h, edge]; host numpy arrays with a leading batch dim, or device
        pointers (DevBuf / torch tensors / ints) with n, conf_shape (K,gh,gw) and edge_shape (E,nh,nw,gh,gw) given."""
        if not on_device:
            tensors = [np.ascontiguousarray(t, np.float32) for t in tensors]
            n = tensors[0].shape[0]
            conf_shape, edge_shape = tensors[0].shape[1:], tensors[6].shape[1:]
            ptrs = (C.c_void_p * 7)(*[t.ctypes.data for t in tensors])
        else:
            ptrs = (C.c_void_p * 7)(*[as_ptr(t).value for t in tensors])
        cs, es = (C.c_int * 3)(*conf_shape), (C.c_int * 5)(*edge_shape)
        check(lib().hp_ppn_process_batch(self._h, n, ptrs, cs, es, int(on_device), self._out, self.cap, self._n))
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[f * self.cap: f * self.cap + self._n[f]].copy() for f in range(n)]

    def process(self, tensors):
        return self.process_batch([np.asarray(t)[None] for t in tensors])[0]

    def enqueue(self, dev_tensors, n: int, conf_shape, edge_shape, stream=None):
        """Asynchronous half: launch on ``stream`` (None = the parser's own stream) and return at once."""
        ptrs = (C.c_void_p * 7)(*[as_ptr(t).value for t in dev_tensors])
        cs, es = (C.c_int * 3)(*conf_shape), (C.c_int * 5)(*edge_shape)
        check(lib().hp_ppn_enqueue(self._h, n, ptrs, cs, es, C.c_void_p(stream) if stream else None))
        self._pending = n

    def decode_flags(self, n: int) -> np.ndarray:
        """Per frame of the last collected batch: 0 = assembled by ppn_assemble_kernel, > 0 = why the device tail declined it, -1 = host."""
        fl = (C.c_int * n)()
        check(lib().hp_ppn_decode_flags(self._h, fl, n))
        return np.array(fl[:n])

    def collect(self):
        n = self._pending
        check(lib().hp_ppn_collect(self._h, self._out, self.cap, self._n))
        self._pending = 0
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[f * self.cap: f * self.cap + self._n[f]].copy() for f in range(n)]


class PifPaf:
    """``hyperpose::parser::pifpaf(h, w, thresh=0.1)`` (reference include/hyperpose/operator/parser/pifpaf.hpp:8-26).
    ``process(paf, pif)`` takes the tensors in the order of the reference's .cpp (src/pifpaf.cpp:7)."""

    def __init__(self, h: int, w: int, thresh: float = 0.1, max_batch: int = 8, cap_per_frame: int = 128):
        self._h = C.c_void_p()
        self.max_batch, self.cap = int(max_batch), int(cap_per_frame)
        check(lib().hp_pifpaf_create(C.byref(self._h), int(h), int(w), C.c_float(thresh), self.max_batch))
        self._out = (Human * (self.max_batch * self.cap))()
        self._n = (C.c_int * self.max_batch)()

    def close(self):
        if self._h:
            lib().hp_pifpaf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_batch(self, paf, pif, on_device: bool = False, n: int = None, fh: int = None, fw: int = None):
        if not on_device:
            paf = np.ascontiguousarray(paf, np.float32)
            pif = np.ascontiguousarray(pif, np.float32)
            n, fh, fw = pif.shape[0], pif.shape[-2], pif.shape[-1]
            a, b = paf.ctypes.data_as(_FP), pif.ctypes.data_as(_FP)
        else:
            a, b = as_ptr(paf), as_ptr(pif)
        check(lib().hp_pifpaf_process_batch(self._h, n, a, b, fh, fw, int(on_device), self._out, self.cap, self._n))
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[f * self.cap: f * self.cap + self._n[f]].copy() for f in range(n)]

    def process(self, paf, pif):
        return self.process_batch(np.asarray(paf)[None], np.asarray(pif)[None])[0]

    def decode_flags(self, n: int):
        """Per frame of the last batch: 0 = decoded by the device kernel, -1 = host tail by configuration (HP_PIFPAF_HOST_TAIL=1),
        > 0 = the reason the device decoder handed the frame to the host tail (include/hp_hip.h)."""
        flags = (C.c_int * n)()
        check(lib().hp_pifpaf_decode_flags(self._h, flags, n))
        return list(flags)

    def enqueue(self, dev_paf, dev_pif, n: int, fh: int, fw: int, stream=None):
        check(lib().hp_pifpaf_enqueue(self._h, n, as_ptr(dev_paf), as_ptr(dev_pif), fh, fw, C.c_void_p(stream) if stream else None))
        self._pending = n

    def collect(self):
        n = self._pending
        check(lib().hp_pifpaf_collect(self._h, self._out, self.cap, self._n))
        self._pending = 0
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[f * self.cap: f * self.cap + self._n[f]].copy() for f in range(n)]
