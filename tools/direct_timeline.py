"""Block timelines (s_memtime = shader cycles) of conv32_direct_kernel on every layer of LW-OpenPose @ 368x432x8 - run on the GPU box:
    python tools/direct_timeline.py [f32|f32s]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["HP_DIRECT_DBG"] = "1"
import numpy as np  # noqa: E402

from hyperpose_amd import _lib  # noqa: E402
from hyperpose_amd.engine import Engine, Model  # noqa: E402

_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
eng = Engine.from_model(m, m.init_weights(1), max_batch=8, dtype=sys.argv[1] if len(sys.argv) > 1 else "f32s")
eng.set_graph(False)
fr = np.random.default_rng(1).integers(0, 256, (8, 368, 432, 3), dtype=np.uint8)
eng.inference(fr)
eng.inference(fr)
