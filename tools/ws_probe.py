"""Timing probe (GPU box): ResNet's 256 -> 1024 expansion at configs[4]'s size, one layer, repeated - which part of conv1x1_ws_kernel /
conv1x1_big_kernel the time belongs to (HP_WS_MODE bit 0: no stores, bit 1: no shortcut, bit 2: no MFMAs; HP_NO_WS1X1=1: the non-persistent kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperpose_amd import _lib
from hyperpose_amd import engine as E
_lib.init(0)
B, H, W = 64, 49, 49
rng = np.random.default_rng(0)
layers, blob = [], []
def alloc(n, std):
    off = sum(len(x) for x in blob)
    blob.append(rng.normal(0, std, n).astype(np.float32))
    return off
# first conv 3 -> 256 (stride 1 on a 49x49 input), shortcut source 3 -> 1024 (1x1), then the expansion 256 -> 1024 + shortcut, relu
w0, b0 = alloc(256 * 27, 0.2), alloc(256, 0.1)
layers.append(E.make_layer(E.OP_CONV, 0, 1, 3, 256, 3, 1, 1, E.ACT_RELU, w_off=w0, b_off=b0))
w1, b1 = alloc(1024 * 3, 0.2), alloc(1024, 0.1)
layers.append(E.make_layer(E.OP_CONV, 0, 2, 3, 1024, 1, 1, 1, E.ACT_RELU, w_off=w1, b_off=b1))
w2, b2 = alloc(1024 * 256, 0.05), alloc(1024, 0.1)
layers.append(E.make_layer(E.OP_CONV, 1, 3, 256, 1024, 1, 1, 1, E.ACT_RELU, res=2, res_before_act=1, w_off=w2, b_off=b2))
w3, b3 = alloc(8 * 1024, 0.05), alloc(8, 0.1)
layers.append(E.make_layer(E.OP_CONV, 3, 4, 1024, 8, 1, 1, 1, E.ACT_NONE, w_off=w3, b_off=b3))
o = E.OutputDesc()
o.name, o.tensor, o.coff, o.channels = b"y", 4, 0, 8
eng = E.Engine(layers, [o], np.concatenate(blob), W, H, B)
for q in eng.profile(B, 20):
    if q["layer"] == 2:
        print(f"mode {os.environ.get('HP_WS_MODE', '0')} nows {os.environ.get('HP_NO_WS1X1', '0')}: tile {q['tile']} {q['ms'] * 1e3:.1f} us")
