"""How the end-to-end step time of bench.py's headline splits when several pipes overlap: ms per step for 1..8 pipes with the
engine only, the parser only, and both (same Pipe objects and loop as bench.py).  Usage: python tools/pipe_sweep.py [config] [f16|f32|f32s]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from hyperpose_amd import _lib
from hyperpose_amd.engine import Model


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cfg = bench.config(idx, sys.argv[2] if len(sys.argv) > 2 else "f16")
    _lib.init(0)
    batch = cfg["batch"]
    model = Model(cfg["arch"], cfg["w"], cfg["h"])
    w = model.init_weights(cfg["seed"])
    frames, maps = bench.synth_inputs(cfg, batch, 0)
    frames_dev = _lib.DevBuf.from_numpy(frames)
    inj = [_lib.DevBuf.from_numpy(m) for m in maps]
    pipes_all = [bench.Pipe(cfg, model, w, inj, batch) for _ in range(8)]

    def rate(pipes, eng, par, min_s=0.5):
        def loop(n):
            for i in range(n):
                p = pipes[i % len(pipes)]
                p.collect()
                p.submit(frames_dev, True, engine=eng, parser=par)
            for p in pipes:
                p.collect()
        t = time.perf_counter()
        while time.perf_counter() - t < 0.25:
            loop(len(pipes) * 4)
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < min_s:
            loop(len(pipes) * 8)
            n += len(pipes) * 8
        return (time.perf_counter() - t0) / n * 1e3

    print("pipes  engine  parser  both   (ms per step of %d frames)" % batch)
    for n in (1, 2, 3, 4, 6, 8):
        ps = pipes_all[:n]
        e, p, b = rate(ps, True, False), rate(ps, False, True), rate(ps, True, True)
        print(f"{n:5d} {e:7.4f} {p:7.4f} {b:7.4f}   -> {batch / b * 1e3:8.0f} fps")


if __name__ == "__main__":
    main()
