"""``hyperpose::make_stream(engine, parser)`` on the GPU (reference include/hyperpose/stream/stream.hpp:119-145):
host frames of any size in, humans out, in submission order; see ``hp_pipeline_*`` in include/hp_hip.h."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import HUMAN_DTYPE, Human, check, lib
from .engine import _DTYPES, EngineDesc, Layer, OutputDesc


class ParserDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("thresh", C.c_float * 3), ("res_w", C.c_int32), ("res_h", C.c_int32)]


PARSER_PAF, PARSER_PPN, PARSER_PIFPAF = 0, 1, 2


class Pipeline:
    """``parser`` selects the hyperpose::parser the stream ends in: "paf" (conf_thresh, paf_thresh), "ppn" (thresholds =
    (point, limb, nms), default 0.10 / 0.05 / 0.3) or "pifpaf" (thresholds = (thresh,), default 0.1)."""

    def __init__(self, model, weights: np.ndarray, max_batch: int = 8, n_pipes: int = 4, keep_ratio: bool = False,
                 conf_thresh: float = 0.05, paf_thresh: float = 0.05, max_frame_wh=(1920, 1080), factor: float = 1.0 / 255,
                 flip_rgb: bool = True, cap_per_frame: int = 128, parser: str = "paf", thresholds=None, dtype="f16"):
        self._h = C.c_void_p()
        weights = np.ascontiguousarray(weights, np.float32)
        larr = (Layer * len(model.layers))(*model.layers)
        oarr = (OutputDesc * len(model.outputs))(*model.outputs)
        d = EngineDesc(model.in_w, model.in_h, max_batch, factor, int(flip_rgb), (C.c_float * 3)(*model.mean),
                       (C.c_float * 3)(*model.inv_std), larr, len(model.layers), oarr, len(model.outputs),
                       weights.ctypes.data_as(C.POINTER(C.c_float)), weights.size, _DTYPES[dtype])
        kind = {"paf": PARSER_PAF, "ppn": PARSER_PPN, "pifpaf": PARSER_PIFPAF}[parser]
        th = thresholds if thresholds is not None else {PARSER_PAF: (conf_thresh, paf_thresh, 0.0), PARSER_PPN: (0.10, 0.05, 0.3),
                                                        PARSER_PIFPAF: (0.1, 0.0, 0.0)}[kind]
        th = tuple(th) + (0.0,) * (3 - len(th))
        pd = ParserDesc(kind, (C.c_float * 3)(*th), -1, -1)
        check(lib().hp_pipeline_create_ex(C.byref(self._h), C.byref(d), C.byref(pd), n_pipes, int(keep_ratio),
                                          C.c_size_t(max_frame_wh[0] * max_frame_wh[1] * 3)))
        self.max_batch, self.n_pipes, self.cap = max_batch, n_pipes, cap_per_frame
        self._out = (Human * (max_batch * cap_per_frame))()
        self._n = (C.c_int * max_batch)()

    def close(self):
        if self._h:
            lib().hp_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def in_flight(self) -> int:
        return lib().hp_pipeline_in_flight(self._h)

    def submit(self, frames) -> None:
        """frames: list of [h, w, 3] uint8 BGR arrays (any sizes), at most max_batch."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        n = len(frames)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[f.ctypes.data_as(C.POINTER(C.c_uint8)) for f in frames])
        ws = (C.c_int * n)(*[f.shape[1] for f in frames])
        hs = (C.c_int * n)(*[f.shape[0] for f in frames])
        check(lib().hp_pipeline_submit(self._h, ptrs, ws, hs, n))
        self._keep = frames  # the async copies read the arrays until the batch is collected

    def submit_ptrs(self, ptrs, ws, hs, n: int) -> None:
        """Pre-marshalled form for hot loops (pinned frames allocated with hp_malloc_host)."""
        check(lib().hp_pipeline_submit(self._h, ptrs, ws, hs, n))

    def collect(self):
        """Humans of the oldest batch in flight: list (per frame) of Human structure arrays."""
        nf = C.c_int(0)
        check(lib().hp_pipeline_collect(self._h, self._out, self.cap, self._n, C.byref(nf)))
        arr = np.frombuffer(self._out, dtype=HUMAN_DTYPE)
        return [arr[i * self.cap: i * self.cap + min(self._n[i], self.cap)].copy() for i in range(nf.value)]
