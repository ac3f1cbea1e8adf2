HP_BN_DBG=1 python - 2>&1 <<'PY' | grep -A1 "bottleneck layer" | head -30
import sys; sys.path.insert(0, '.')
from hyperpose_amd import _lib
from hyperpose_amd.engine import Engine, Model
_lib.init(0)
m = Model("pifpaf_resnet50", 385, 385)
eng = Engine.from_model(m, m.init_weights(1), max_batch=64)
eng.profile(64, iters=1)
PY
