"""Host-side mirror of ``hyperpose::dnn::tensorrt`` (reference include/hyperpose/operator/dnn/tensorrt.hpp:33-141)
over the C ABI: ``Engine.inference(frames)`` has the meaning of ``tensorrt::inference(std::vector<cv::Mat>)`` for
network-sized frames; outputs are returned per image, ordered by tensor name (src/tensorrt.cpp:405).  ``Model``
wraps the built-in topology builders.  All arithmetic runs in libhp_hip.so; this file only marshals.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_ptr, check, lib

OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_UPSAMPLE = 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY, ACT_PRELU, ACT_SIGMOID, ACT_SOFTPLUS = range(7)


class Layer(C.Structure):
    _fields_ = [("op", C.c_int32), ("in_", C.c_int32), ("in_coff", C.c_int32), ("res", C.c_int32),
                ("res_before_act", C.c_int32), ("out", C.c_int32), ("out_coff", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32),
                ("act", C.c_int32), ("act_param", C.c_float), ("w_off", C.c_int64), ("b_off", C.c_int64),
                ("alpha_off", C.c_int64), ("pad_explicit", C.c_int32), ("pad", C.c_int32 * 4)]


class OutputDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("tensor", C.c_int32), ("coff", C.c_int32), ("channels", C.c_int32),
                ("act", C.c_int32), ("shuffle", C.c_int32), ("group", C.c_int32), ("sigmoid_mask", C.c_uint32),
                ("softplus_mask", C.c_uint32), ("out_h", C.c_int32), ("out_w", C.c_int32), ("scale", C.c_float),
                ("grid", C.c_int32)]


class EngineDesc(C.Structure):
    _fields_ = [("in_w", C.c_int32), ("in_h", C.c_int32), ("max_batch", C.c_int32), ("factor", C.c_double),
                ("flip_rb", C.c_int32), ("mean", C.c_float * 3), ("inv_std", C.c_float * 3),
                ("layers", C.POINTER(Layer)), ("n_layers", C.c_int32), ("outputs", C.POINTER(OutputDesc)),
                ("n_outputs", C.c_int32), ("weights", C.POINTER(C.c_float)), ("n_weights", C.c_size_t),
                ("dtype", C.c_int32)]


# HP_DTYPE_*: data_type::kHALF / data_type::kFLOAT of the reference's engine; F32S = the kFLOAT engine with the dense layers' products
# formed as three exact fp16 x fp16 products on the fp16 matrix pipe (csrc/conv32_direct.hip), opt-in
DTYPE_F16, DTYPE_F32, DTYPE_F32S = 0, 1, 2
_DTYPES = {"f16": DTYPE_F16, "fp16": DTYPE_F16, "half": DTYPE_F16, DTYPE_F16: DTYPE_F16,
           "f32": DTYPE_F32, "fp32": DTYPE_F32, "float": DTYPE_F32, DTYPE_F32: DTYPE_F32,
           "f32s": DTYPE_F32S, "f32_split": DTYPE_F32S, DTYPE_F32S: DTYPE_F32S}


class LayerTime(C.Structure):
    _fields_ = [("layer", C.c_int32), ("op", C.c_int32), ("tile", C.c_int32), ("ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double)]


def make_layer(op, in_, out, cin, cout, k=1, stride=1, dil=1, act=ACT_NONE, in_coff=0, out_coff=0, res=-1,
               res_before_act=0, w_off=-1, b_off=-1, alpha_off=-1, act_param=0.0, pads=None) -> Layer:
    """``pads`` = (top, left, bottom, right) as in ONNX; None = TF "SAME"."""
    L = Layer(op, in_, in_coff, res, res_before_act, out, out_coff, cin, cout, k, k, stride, dil, act, act_param,
              w_off, b_off, alpha_off)
    if pads is not None:
        L.pad_explicit = 1
        L.pad[:] = [int(v) for v in pads]
    return L


class Model:
    """A built-in topology (``hp_model_*``): layer list, outputs, synthetic weights."""

    def __init__(self, arch: str, in_w: int, in_h: int):
        self._h = C.c_void_p()
        self.arch, self.in_w, self.in_h = arch, in_w, in_h
        check(lib().hp_model_build(C.byref(self._h), arch.encode(), in_w, in_h))
        self._read()

    @classmethod
    def from_onnx(cls, model, in_w: int = 0, in_h: int = 0) -> "Model":
        """``hyperpose::dnn::onnx{path}`` (include/hyperpose/utility/model.hpp:23-25): a path or the file's bytes.  The input
        size is the caller's, as in ``tensorrt(onnx, cv::Size, ...)``; 0, 0 takes the static size stored in the graph.  The
        imported weights are ``self.weights``."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        if isinstance(model, (bytes, bytearray, memoryview)):
            buf = bytes(model)
            check(lib().hp_model_from_onnx(C.byref(self._h), buf, C.c_size_t(len(buf)), int(in_w), int(in_h)))
            self.arch = "onnx"
        else:
            check(lib().hp_model_from_onnx_file(C.byref(self._h), str(model).encode(), int(in_w), int(in_h)))
            self.arch = "onnx:" + str(model)
        w, h = C.c_int(), C.c_int()
        check(lib().hp_model_input_size(self._h, C.byref(w), C.byref(h)))
        self.in_w, self.in_h = w.value, h.value
        self._read()
        blob, n = C.POINTER(C.c_float)(), C.c_size_t()
        check(lib().hp_model_weights(self._h, C.byref(blob), C.byref(n)))
        self.weights = np.ctypeslib.as_array(blob, shape=(n.value,)).copy()
        return self

    def _read(self):
        lp, n = C.POINTER(Layer)(), C.c_int(0)
        check(lib().hp_model_layers(self._h, C.byref(lp), C.byref(n)))
        self.layers = [lp[i] for i in range(n.value)]
        op, m = C.POINTER(OutputDesc)(), C.c_int(0)
        check(lib().hp_model_outputs(self._h, C.byref(op), C.byref(m)))
        self.outputs = [op[i] for i in range(m.value)]
        lib().hp_model_num_weights.restype = C.c_size_t
        self.n_weights = lib().hp_model_num_weights(self._h)
        lib().hp_model_flops_per_frame.restype = C.c_double
        self.flops_per_frame = lib().hp_model_flops_per_frame(self._h)
        mean, inv_std = (C.c_float * 3)(), (C.c_float * 3)()
        check(lib().hp_model_preproc(self._h, mean, inv_std))
        self.mean, self.inv_std = list(mean), list(inv_std)

    @staticmethod
    def archs():
        lib().hp_model_archs.restype = C.c_char_p
        return lib().hp_model_archs().decode().split(",")

    def init_weights(self, seed: int = 20240) -> np.ndarray:
        blob = np.zeros(self.n_weights, np.float32)
        check(lib().hp_model_init_weights(self._h, C.c_uint64(seed), blob.ctypes.data_as(C.POINTER(C.c_float)),
                                          C.c_size_t(blob.size)))
        return blob

    def close(self):
        if self._h:
            lib().hp_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """``hyperpose::dnn`` engine: (layers, outputs, weights) -> device-resident fp32 NCHW feature maps."""

    def __init__(self, layers, outputs, weights: np.ndarray, in_w: int, in_h: int, max_batch: int = 8,
                 factor: float = 1.0 / 255, flip_rgb: bool = True, mean=(0, 0, 0), inv_std=(1, 1, 1), dtype="f16"):
        """``dtype``: "f16" (fp16 storage, fp32 accumulation: the fast path, ``data_type::kHALF``) or "f32" (fp32 storage and
        arithmetic: what ``data_type::kFLOAT``, the reference's default, promises)."""
        self._h = C.c_void_p()
        self.dtype = _DTYPES[dtype]
        weights = np.ascontiguousarray(weights, np.float32)
        larr = (Layer * len(layers))(*layers)
        oarr = (OutputDesc * len(outputs))(*outputs)
        d = EngineDesc(in_w, in_h, max_batch, factor, int(flip_rgb), (C.c_float * 3)(*mean), (C.c_float * 3)(*inv_std),
                       larr, len(layers), oarr, len(outputs), weights.ctypes.data_as(C.POINTER(C.c_float)), weights.size,
                       self.dtype)
        check(lib().hp_engine_create(C.byref(self._h), C.byref(d)))
        self.in_w, self.in_h, self.max_batch = in_w, in_h, max_batch
        lib().hp_engine_stream.restype = C.c_void_p
        self.stream = lib().hp_engine_stream(self._h)
        self.outputs = []
        for i in range(lib().hp_engine_num_outputs(self._h)):
            name, shape, dev = C.c_char_p(), (C.c_int * 3)(), C.POINTER(C.c_float)()
            check(lib().hp_engine_output(self._h, i, C.byref(name), shape, C.byref(dev)))
            self.outputs.append((name.value.decode(), tuple(shape), C.cast(dev, C.c_void_p).value))

    @classmethod
    def from_model(cls, model: Model, weights: np.ndarray, max_batch: int = 8, factor: float = 1.0 / 255,
                   flip_rgb: bool = True, dtype="f16") -> "Engine":
        return cls(model.layers, model.outputs, weights, model.in_w, model.in_h, max_batch, factor, flip_rgb,
                   model.mean, model.inv_std, dtype)

    def _adopt(self, handle, max_batch: int):
        self._h = handle
        w, h = C.c_int(), C.c_int()
        check(lib().hp_engine_input_size(self._h, C.byref(w), C.byref(h)))
        self.in_w, self.in_h, self.max_batch = w.value, h.value, lib().hp_engine_max_batch(self._h)
        self.dtype = lib().hp_engine_dtype(self._h)
        lib().hp_engine_stream.restype = C.c_void_p
        self.stream = lib().hp_engine_stream(self._h)
        self.outputs = []
        for i in range(lib().hp_engine_num_outputs(self._h)):
            name, shape, dev = C.c_char_p(), (C.c_int * 3)(), C.POINTER(C.c_float)()
            check(lib().hp_engine_output(self._h, i, C.byref(name), shape, C.byref(dev)))
            self.outputs.append((name.value.decode(), tuple(shape), C.cast(dev, C.c_void_p).value))

    def save(self, path: str) -> None:
        """``tensorrt::save`` (src/tensorrt.cpp:463-471): topology + pre-processing + weights in one file."""
        check(lib().hp_engine_save(self._h, path.encode()))

    @classmethod
    def load(cls, path: str, max_batch: int = 0) -> "Engine":
        """``tensorrt(tensorrt_serialized{path}, ...)`` (include/hyperpose/utility/model.hpp:27-32)."""
        self = cls.__new__(cls)
        handle = C.c_void_p()
        check(lib().hp_engine_load(C.byref(handle), path.encode(), int(max_batch)))
        self._adopt(handle, max_batch)
        return self

    @property
    def split_fallbacks(self) -> int:
        """HP_DTYPE_F32S engines: 1 once an activation beyond fp16's range sent the engine back to the fp32 matrix pipe."""
        return int(lib().hp_engine_split_fallbacks(self._h))

    @property
    def device_bytes(self) -> dict:
        """HBM the engine holds for its max_batch (hp_engine_device_bytes): activation tensors, packed weights, fp32 network outputs."""
        b = (C.c_uint64 * 3)()
        check(lib().hp_engine_device_bytes(self._h, b))
        return {"activations": int(b[0]), "weights": int(b[1]), "outputs": int(b[2]), "total": int(b[0] + b[1] + b[2])}

    @property
    def arena_info(self) -> dict:
        """The activation arena of an fp32 engine (hp_engine_arena_info): buffers, tensors living in them, bytes without re-use."""
        b = (C.c_uint64 * 3)()
        check(lib().hp_engine_arena_info(self._h, b))
        return {"buffers": int(b[0]), "tensors": int(b[1]), "bytes_without_reuse": int(b[2])}

    def close(self):
        if self._h:
            lib().hp_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def input_size(self):
        return (self.in_w, self.in_h)

    def set_graph(self, enable: bool):
        check(lib().hp_engine_set_graph(self._h, int(enable)))

    # ---- asynchronous device-resident path (bench, pipelines)
    def set_concurrency(self, parts: int):
        """2: a batch runs as two half-batches side by side on two internal streams (hp_engine_set_concurrency; fp32 engines) - for a
        caller with one batch in flight, like the reference's synchronous ``tensorrt::inference``.  Bit-identical outputs."""
        check(lib().hp_engine_set_concurrency(self._h, int(parts)))

    @property
    def concurrency(self) -> int:
        return int(lib().hp_engine_concurrency(self._h))

    def enqueue_u8(self, dev_frames, n: int, stream=None):
        check(lib().hp_engine_infer_u8(self._h, as_ptr(dev_frames), n, 1, C.c_void_p(stream) if stream else None))

    def synchronize(self):
        check(lib().hp_engine_synchronize(self._h))

    def output_to_host(self, i: int, n: int) -> np.ndarray:
        name, shape, _ = self.outputs[i]
        out = np.empty((n,) + shape, np.float32)
        check(lib().hp_engine_output_to_host(self._h, i, n, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # ---- reference-shaped API
    def inference(self, frames: np.ndarray):
        """u8 HWC BGR frames [n,h,w,3] (network-sized) -> list (per image) of list of (name, fp32 array [C,H,W])."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n = frames.shape[0]
        if n > self.max_batch:  # src/tensorrt.cpp:439-443
            raise ValueError(f"Input batch size overflow: Yours@{n} Max@{self.max_batch}")
        assert frames.shape[1:] == (self.in_h, self.in_w, 3)
        check(lib().hp_engine_infer_u8(self._h, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, 0, None))
        outs = [self.output_to_host(i, n) for i in range(len(self.outputs))]
        return [[(self.outputs[i][0], outs[i][b]) for i in range(len(self.outputs))] for b in range(n)]

    def inference_f32(self, nchw: np.ndarray):
        nchw = np.ascontiguousarray(nchw, np.float32)
        n = nchw.shape[0]
        if n > self.max_batch:
            raise ValueError(f"Input batch size overflow: Yours@{n} Max@{self.max_batch}")
        check(lib().hp_engine_infer_f32(self._h, nchw.ctypes.data_as(C.POINTER(C.c_float)), n, 0, None))
        outs = [self.output_to_host(i, n) for i in range(len(self.outputs))]
        return [[(self.outputs[i][0], outs[i][b]) for i in range(len(self.outputs))] for b in range(n)]

    def debug_tensor(self, tensor: int, n: int) -> np.ndarray:
        shape = (C.c_int * 3)()
        check(lib().hp_engine_debug_tensor(self._h, tensor, n, None, shape))
        out = np.empty((n,) + tuple(shape), np.float32)
        check(lib().hp_engine_debug_tensor(self._h, tensor, n, out.ctypes.data_as(C.POINTER(C.c_float)), shape))
        return out

    def profile(self, n: int, iters: int = 10, in_sequence: bool = False, pair: "Engine | None" = None):
        """Per-step device times: each step launched back to back (default) or the whole schedule in order with events in
        between (``in_sequence``: the cache state of a real inference; agrees with rocprofv3's per-kernel averages), or - ``pair`` =
        a second engine of the same model - alternately on the two engines' streams (machine time per launch when two pipes overlap)."""
        cap = 1024
        buf = (LayerTime * cap)()
        cnt = C.c_int(0)
        if pair is not None:
            check(lib().hp_engine_profile_pair(self._h, pair._h, n, iters, buf, cap, C.byref(cnt)))
        else:
            fn = lib().hp_engine_profile_sequence if in_sequence else lib().hp_engine_profile
            check(fn(self._h, n, iters, buf, cap, C.byref(cnt)))
        return [dict(layer=b.layer, op=b.op, tile=b.tile, ms=b.ms, flops=b.flops, bytes=b.bytes) for b in buf[:cnt.value]]
