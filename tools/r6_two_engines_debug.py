"""two independent engines on two streams, same frames: are their outputs those of one engine alone?  (python tools/r6_two_engines_debug.py f32s|f32|f16)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hyperpose_amd import _lib, synth
from hyperpose_amd import engine as E
_lib.init(0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
m = E.Model("lw_openpose_mobilenet", 432, 368)
w = m.init_weights(5)
n = 8
engs = [E.Engine.from_model(m, w, max_batch=n, dtype=dtype) for _ in range(2)]
fr = synth.images_u8(synth.rng_for(12), n, 368, 432)
dev = _lib.DevBuf.from_numpy(fr)
def outs(e):
    return [e.output_to_host(i, n) for i in range(len(e.outputs))]
engs[0].enqueue_u8(dev, n); engs[0].synchronize()
one = outs(engs[0])
hits = 0
for rep in range(60):
    for e in engs:
        e.enqueue_u8(dev, n)
    for e in engs:
        e.synchronize()
    for k, e in enumerate(engs):
        r = outs(e)
        bad = [(i, sorted(set(np.argwhere(a != b)[:, 0]))) for i, (a, b) in enumerate(zip(one, r)) if not np.array_equal(a, b)]
        if bad:
            hits += 1
            print("rep", rep, "engine", k, "outputs differ (output, frames):", bad)
print(dtype, "two engines side by side: mismatching (rep, engine) pairs:", hits, "of 120")
