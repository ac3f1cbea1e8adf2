"""tools/parser_dnn_probe.py - the PAF parser on the network's OWN maps (random weights: dense noise) against the injected maps with people:
peaks / candidates / humans per frame and ms per batch (run on the GPU box; under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hyperpose_amd import _lib, synth
from hyperpose_amd.engine import Engine, Model
from hyperpose_amd.parser import Paf
_lib.init(0)
m = Model("lw_openpose_mobilenet", 432, 368)
w = m.init_weights(20241)
eng = Engine.from_model(m, w, max_batch=8, dtype="f32")
rng = synth.rng_for(1, salt=0)
frames = synth.images_u8(rng, 8, 368, 432)
got = eng.inference(frames)
conf = np.stack([g[0][1] for g in got]); paf = np.stack([g[1][1] for g in got])
cinj, pinj, _ = synth.paf_maps(rng, 8, 46, 54, people=(1, 2, 4, 8, 16, 3, 5, 6))
for name, c, p in (("dnn-output", conf, paf), ("injected", cinj, pinj)):
    par = Paf(max_batch=8)
    hs = par.process_batch(c, p)
    pk = [len(par.debug_peaks(b, cap=65536)) for b in range(8)]
    print(f"{name}: conf range [{c.min():.3f}, {c.max():.3f}] mean {c.mean():.3f}; fraction of conf > 0.05: {(c[:, :18] > 0.05).mean():.3f}; peaks per frame {pk}; humans per frame {[len(h) for h in hs]}")
    dc, dp = _lib.DevBuf.from_numpy(c), _lib.DevBuf.from_numpy(p)
    for _ in range(20):
        par.enqueue(dc, dp, 8, c.shape[1:], p.shape[1:]); par.collect()
    t0 = time.perf_counter(); n = 200
    for _ in range(n):
        par.enqueue(dc, dp, 8, c.shape[1:], p.shape[1:]); par.collect()
    print(f"   {(time.perf_counter() - t0) / n * 1e3:.4f} ms per batch of 8 (enqueue + collect)")
