"""Block timelines (s_memtime = shader cycles) of the fp32 engines' kernels on every layer of a model (default LW-OpenPose @ 368x432x8) - run on
the GPU box:    python tools/direct_timeline.py [f32|f32s] [model w h batch]        (HP_DIRECT_DBG_MINCIN=64: conv32 kernels from 64 input channels)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["HP_DIRECT_DBG"] = "1"
import numpy as np  # noqa: E402

from hyperpose_amd import _lib  # noqa: E402
from hyperpose_amd.engine import Engine, Model  # noqa: E402

_lib.init(0)
name, w, h, nb = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ("lw_openpose_mobilenet", 432, 368, 8)
m = Model(name, w, h)
eng = Engine.from_model(m, m.init_weights(1), max_batch=nb, dtype=sys.argv[1] if len(sys.argv) > 1 else "f32s")
eng.set_graph(False)
fr = np.random.default_rng(1).integers(0, 256, (nb, h, w, 3), dtype=np.uint8)
eng.inference(fr)
eng.inference(fr)
