"""Where does the end-to-end time go?  engine-only / parser-only / both, with host-side time per call (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperpose_amd import _lib, synth
from hyperpose_amd.engine import Model

_lib.init(0)
model = Model(bench.ARCH, bench.IN_W, bench.IN_H)
w = model.init_weights(20241)
rng = synth.rng_for(1, salt=0)
frames = synth.images_u8(rng, bench.BATCH, bench.IN_H, bench.IN_W)
conf, paf, _ = synth.paf_maps(rng, bench.BATCH, bench.IN_H // 8, bench.IN_W // 8, people=(1, 2, 4, 8, 16, 3, 5, 6))
fd, cd, pd = _lib.DevBuf.from_numpy(frames), _lib.DevBuf.from_numpy(conf), _lib.DevBuf.from_numpy(paf)
B = bench.BATCH
for npipes in (1, 2, 4, 8):
    pipes = [bench.Pipe(model, w, cd, pd) for _ in range(npipes)]
    for mode in ("engine", "parser", "both"):
        t_sub = t_col = 0.0
        def step(p):
            global t_sub, t_col
            t0 = time.perf_counter()
            if p.busy:
                if mode == "engine":
                    p.eng.synchronize()
                else:
                    p.paf.collect()
                p.busy = False
            t1 = time.perf_counter()
            if mode != "parser":
                p.eng.enqueue_u8(fd, B)
            if mode != "engine":
                p.paf.enqueue(cd, pd, B, p.conf_shape, p.paf_shape, stream=p.stream)
            p.busy = True
            t2 = time.perf_counter()
            t_col += t1 - t0
            t_sub += t2 - t1
        for i in range(40):
            step(pipes[i % npipes])
        t_sub = t_col = 0.0
        steps = 400
        t0 = time.perf_counter()
        for i in range(steps):
            step(pipes[i % npipes])
        for p in pipes:
            if p.busy:
                p.eng.synchronize(); p.busy = False
                if mode != "engine":
                    try: p.paf.collect()
                    except Exception: pass
        dt = time.perf_counter() - t0
        print(f"pipes={npipes} {mode:7s}: {B*steps/dt:9.0f} FPS  {dt/steps*1e6:7.1f} us/batch   host submit {t_sub/steps*1e6:6.1f} us  collect(wait+assemble) {t_col/steps*1e6:6.1f} us", flush=True)
    del pipes
