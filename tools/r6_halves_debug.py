import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hyperpose_amd import _lib, synth
from hyperpose_amd import engine as E
_lib.init(0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
mode = sys.argv[2] if len(sys.argv) > 2 else "host"
graph = (sys.argv[3] if len(sys.argv) > 3 else "eager") == "graph"
m = E.Model("lw_openpose_mobilenet", 432, 368)
w = m.init_weights(5)
n = 8
eng = E.Engine.from_model(m, w, max_batch=n, dtype=dtype)
fr = synth.images_u8(synth.rng_for(12), n, 368, 432)
dev = _lib.DevBuf.from_numpy(fr)
def run():
    if mode == "host":
        return eng.inference(fr)
    eng.enqueue_u8(dev, n)
    eng.synchronize()
    return [[(nm, eng.output_to_host(i, n)[b]) for i, (nm, _, _) in enumerate(eng.outputs)] for b in range(n)]
one = run()
base1 = eng.debug_tensor(1, n)
eng.set_graph(graph)
eng.set_concurrency(2)
hits = 0
for rep in range(int(os.environ.get("REPS", "60"))):
    r = run()
    bad = [(b, nm) for b in range(n) for (nm, x), (_, y) in zip(one[b], r[b]) if not np.array_equal(x, y)]
    if bad:
        hits += 1
        cur = eng.debug_tensor(1, n)
        d = np.argwhere(cur != base1)
        print("rep", rep, "outputs differ:", bad[:4], "| tensor 1 differing elements", len(d), "frames", sorted(set(d[:, 0])) if len(d) else [])
        for f, c, y, x in d[:4 if hits > 1 else 14]:
            print("    frame", f, "ch", c, "y", y, "x", x, "good", base1[f, c, y, x], "now", cur[f, c, y, x], "| same place 4 frames earlier", base1[f - 4, c, y, x])
print(dtype, mode, "graph" if graph else "eager", "mismatching runs:", hits, "of", os.environ.get("REPS", "60"))
