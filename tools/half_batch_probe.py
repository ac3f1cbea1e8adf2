"""tools/half_batch_probe.py — does ONE synchronous call get faster when its batch runs as two half-batches on two streams?  

Run on the GPU box:   python tools/half_batch_probe.py [f32|f16]

A: one engine, batch 8, one stream: enqueue + synchronize, per call.
B: two engines of batch 4 on their own streams, both enqueued, both synchronized: the same eight frames per "call".
C: as B with four engines of batch 2.
What the kernels of two half-batches share is the chip, not their phases: a store-bound launch of one half runs under an MFMA-bound launch of the other.
"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from hyperpose_amd import _lib  # noqa: E402
from hyperpose_amd.engine import Engine, Model  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
w = m.init_weights(1)
frames = np.random.default_rng(1).integers(0, 256, (8, 368, 432, 3), dtype=np.uint8)


def rate(parts):
    n = 8 // parts
    engs = [Engine.from_model(m, w, max_batch=n, dtype=dtype) for _ in range(parts)]
    devs = [_lib.DevBuf.from_numpy(frames[i * n:(i + 1) * n]) for i in range(parts)]

    def call():
        for e, d in zip(engs, devs):
            e.enqueue_u8(d, n)
        for e in engs:
            e.synchronize()

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        call()
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        call()
        k += 1
    return (time.perf_counter() - t0) / k * 1e6


for parts in (1, 2, 4):
    us = rate(parts)
    print(f"{dtype}: batch 8 as {parts} x {8 // parts} frames on {parts} stream(s): {us:8.1f} us per call of 8 frames -> {8e6 / us:7.0f} frames/s (engine only, synchronous caller)")
