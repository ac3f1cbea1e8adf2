HP_SEP_SLOT=0 HP_SEP_DBG=1 python - 2>&1 <<'PY' | grep "sep layer" | grep "C=512" | tail -4 > gpurun_out/sep_dbg.txt
import sys; sys.path.insert(0, '.')
from hyperpose_amd import _lib
from hyperpose_amd.engine import Engine, Model
_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
eng = Engine.from_model(m, m.init_weights(1), max_batch=8)
eng.profile(8, iters=1)
PY
cat gpurun_out/sep_dbg.txt
python -m pytest tests/test_pipeline_gpu.py -q -k drift -s 2>&1 | grep -E "fp16 engine|passed|failed|classes|Assertion" | tail -5
