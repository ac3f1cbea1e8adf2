"""Throughput of the N-pipe loop under different stream set-ups (stream creation order, CU partitions, HW queues)."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--pipes", type=int, default=4)
ap.add_argument("--engines-first", action="store_true")
ap.add_argument("--modes", default="injected")
ap.add_argument("--tag", default="")
ap.add_argument("--lanes", type=int, default=0)
ap.add_argument("--paf-own-stream", action="store_true")
ap.add_argument("--paf-shared-stream", type=int, default=0)
ap.add_argument("--dummy-mb", type=float, default=0)
ap.add_argument("--dummy-streams", type=int, default=0)
ap.add_argument("--dummy-every", type=int, default=1)
a = ap.parse_args()
import bench
from hyperpose_amd import _lib, synth
from hyperpose_amd.engine import Model, Engine
from hyperpose_amd.parser import Paf
_lib.init(0)
model = Model(bench.ARCH, bench.IN_W, bench.IN_H)
w = model.init_weights(20241)
rng = synth.rng_for(1, salt=0)
frames = synth.images_u8(rng, bench.BATCH, bench.IN_H, bench.IN_W)
conf, paf, _ = synth.paf_maps(rng, bench.BATCH, bench.IN_H // 8, bench.IN_W // 8, people=(1, 2, 4, 8, 16, 3, 5, 6))
fd, cd, pd = _lib.DevBuf.from_numpy(frames), _lib.DevBuf.from_numpy(conf), _lib.DevBuf.from_numpy(paf)
if a.engines_first:
    engs, dummies = [], []
    import ctypes
    for _ in range(a.pipes):
        engs.append(Engine.from_model(model, w, max_batch=bench.BATCH))
        if a.dummy_mb:
            dummies.append(_lib.DevBuf(int(a.dummy_mb * (1 << 20))))
        if len(engs) % a.dummy_every == 0:
            for _ in range(a.dummy_streams):
                dummies.append(Paf(max_batch=1))
    pipes = []
    for e in engs:
        p = bench.Pipe.__new__(bench.Pipe)
        p.eng, p.paf, p.stream, p.conf_dev, p.paf_dev, p.busy = e, Paf(max_batch=bench.BATCH), e.stream, cd, pd, False
        outs = {n: (s, q) for n, s, q in e.outputs}
        p.conf_shape, p.dnn_conf = outs["conf"]; p.paf_shape, p.dnn_paf = outs["paf"]
        pipes.append(p)
else:
    pipes = [bench.Pipe(model, w, cd, pd) for _ in range(a.pipes)]
if a.lanes:
    lane_streams = [pipes[i].eng.stream for i in range(a.lanes)]
    for i, p in enumerate(pipes):
        p.stream = lane_streams[i % a.lanes]
    # engine launches follow the pipe's lane stream
    for p in pipes:
        p.submit = (lambda p: (lambda frames_dev, injected: (
            p.eng.enqueue_u8(frames_dev, bench.BATCH, stream=p.stream),
            p.paf.enqueue(p.conf_dev if injected else p.dnn_conf, p.paf_dev if injected else p.dnn_paf, bench.BATCH, p.conf_shape, p.paf_shape, stream=p.stream),
            setattr(p, "busy", True))))(p)
if a.paf_own_stream:
    for p in pipes:
        def submit(frames_dev, injected, p=p):
            p.eng.enqueue_u8(frames_dev, bench.BATCH)
            p.paf.after(p.eng.stream)
            p.paf.enqueue(p.conf_dev if injected else p.dnn_conf, p.paf_dev if injected else p.dnn_paf, bench.BATCH, p.conf_shape, p.paf_shape)
            p.busy = True
        p.submit = submit
if a.paf_shared_stream:
    import ctypes as C
    from hyperpose_amd._lib import lib, check
    shared = [pipes[i].paf.stream for i in range(a.paf_shared_stream)]
    for k, p in enumerate(pipes):
        def submit(frames_dev, injected, p=p, st=shared[k % len(shared)]):
            p.eng.enqueue_u8(frames_dev, bench.BATCH)
            check(lib().hp_stream_wait_stream(C.c_void_p(st), C.c_void_p(p.eng.stream)))
            p.paf.enqueue(p.conf_dev if injected else p.dnn_conf, p.paf_dev if injected else p.dnn_paf, bench.BATCH, p.conf_shape, p.paf_shape, stream=st)
            p.busy = True
        p.submit = submit
for mode in a.modes.split(","):
    if mode == "engine":
        def loop(n):
            for i in range(n):
                p = pipes[i % len(pipes)]
                if p.busy: p.eng.synchronize()
                p.eng.enqueue_u8(fd, bench.BATCH); p.busy = True
            for p in pipes:
                p.eng.synchronize(); p.busy = False
    else:
        def loop(n, inj=(mode == "injected")):
            bench.run_loop(pipes, fd, n, inj)
    loop(40)
    t0 = time.perf_counter(); loop(300); dt = time.perf_counter() - t0
    env = {k: v for k, v in os.environ.items() if k.startswith(("HP_", "GPU_MAX"))}
    print(f"pipes={a.pipes} shared={a.paf_shared_stream} own={int(a.paf_own_stream)} lanes={a.lanes} ef={int(a.engines_first)} {env} {mode}: {bench.BATCH*300/dt:.0f} FPS {dt/300*1e6:.1f} us/batch", flush=True)
