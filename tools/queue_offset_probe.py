import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from hyperpose_amd import _lib, synth
from hyperpose_amd.engine import Model
import ctypes as C
k = int(sys.argv[1])
torch.cuda.set_device(0); _lib.init(0)
lib = _lib.lib()
# k dummy streams shift the runtime's round-robin queue assignment
dummies = [torch.cuda.Stream() for _ in range(k)]
cfg = bench.config(1, "f16")
model = Model(cfg["arch"], cfg["w"], cfg["h"]); w = model.init_weights(1)
frames, maps = bench.synth_inputs(cfg, 8, 0)
fd = _lib.DevBuf.from_numpy(frames); inj = [_lib.DevBuf.from_numpy(m) for m in maps]
pipes = [bench.Pipe(cfg, model, w, inj, 8) for _ in range(4)]
bench.run_loop(pipes, fd, 200, True); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_loop(pipes, fd, 400, True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fps = 8 * 400 / dt
del pipes
h2d = bench.h2d_inclusive(model, w, cfg, 8, 4, 200)["value"]
print(f"dummy streams {k}: headline {fps:.0f} fps, h2d_inclusive {h2d:.0f}")
