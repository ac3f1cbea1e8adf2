"""Stream front-end geometry over the C ABI: ``cv::resize`` / ``non_scaling_resize`` on device images and
``resume_ratio`` (reference src/stream.cpp:89-103, src/data.cpp:53-69, include/hyperpose/utility/human.hpp:44-58)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import DevBuf, as_ptr, check, lib


def resize(src_dev, sw: int, sh: int, dst_dev, dw: int, dh: int, stream=None, src_stride=None, dst_stride=None) -> None:
    check(lib().hp_resize_u8c3(as_ptr(src_dev), sw, sh, src_stride or sw * 3, as_ptr(dst_dev), dw, dh, dst_stride or dw * 3,
                               C.c_void_p(stream) if stream else None))


def letterbox(src_dev, sw: int, sh: int, dst_dev, dw: int, dh: int, bgcolor=(0, 0, 0), stream=None) -> None:
    check(lib().hp_letterbox_u8c3(as_ptr(src_dev), sw, sh, sw * 3, as_ptr(dst_dev), dw, dh, dw * 3, int(bgcolor[0]), int(bgcolor[1]),
                                  int(bgcolor[2]), C.c_void_p(stream) if stream else None))


def letterbox_inner(sw: int, sh: int, dw: int, dh: int):
    iw, ih = C.c_int(), C.c_int()
    lib().hp_letterbox_inner(sw, sh, dw, dh, C.byref(iw), C.byref(ih))
    return iw.value, ih.value


def resize_host(img: np.ndarray, dw: int, dh: int, keep_ratio: bool = False, bgcolor=(0, 0, 0)) -> np.ndarray:
    """Convenience for tests: host image [h, w, 3] u8 -> device -> resized -> host."""
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw, _ = img.shape
    src = DevBuf.from_numpy(img)
    dst = DevBuf(dw * dh * 3)
    (letterbox(src, sw, sh, dst, dw, dh, bgcolor) if keep_ratio else resize(src, sw, sh, dst, dw, dh))
    check(lib().hp_device_synchronize())
    out = np.empty((dh, dw, 3), np.uint8)
    check(lib().hp_memcpy_d2h(out.ctypes.data_as(C.c_void_p), dst.ptr, C.c_size_t(out.nbytes)))
    return out
