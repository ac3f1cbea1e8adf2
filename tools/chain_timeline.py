import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hyperpose_amd import _lib
from hyperpose_amd.engine import Engine, Model
import numpy as np
os.environ["HP_CHAIN_DBG"] = "1"  # read once when the engine is created
_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
eng = Engine.from_model(m, m.init_weights(1), max_batch=8)
eng.set_graph(False)
fr = np.zeros((8, 368, 432, 3), np.uint8)
eng.inference(fr)
