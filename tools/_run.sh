for a in 0 1 2 3; do
HP_CHAIN_ABL=$a HP_CHAIN_DBG=1 python - 2>&1 <<'PY' | grep "layer 24" | tail -1
import sys; sys.path.insert(0, '.')
from hyperpose_amd import _lib
from hyperpose_amd.engine import Engine, Model
_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
eng = Engine.from_model(m, m.init_weights(1), max_batch=8)
eng.profile(8, iters=1)
PY
done > gpurun_out/chain_abl.txt
cat gpurun_out/chain_abl.txt
