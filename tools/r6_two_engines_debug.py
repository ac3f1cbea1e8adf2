"""two independent engines on two streams, same frames: are their outputs those of one engine alone?  (python tools/r6_two_engines_debug.py f32s|f32|f16)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hyperpose_amd import _lib, synth
from hyperpose_amd import engine as E
_lib.init(0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
dtypes = dtype.split(",") if "," in dtype else [dtype, dtype]   # "f32s,f32": a split engine and an fp32 engine side by side
delay_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0    # the second engine's batch is enqueued this much later
import time
m = E.Model("lw_openpose_mobilenet", 432, 368)
w = m.init_weights(5)
n = 8
engs = [E.Engine.from_model(m, w, max_batch=n, dtype=d) for d in dtypes]
fr = synth.images_u8(synth.rng_for(12), n, 368, 432)
dev = _lib.DevBuf.from_numpy(fr)
def outs(e):
    return [e.output_to_host(i, n) for i in range(len(e.outputs))]
ones, bases = [], []
tens = sorted({L.out for L in m.layers})
for e in engs:
    e.enqueue_u8(dev, n); e.synchronize()
    ones.append(outs(e))
    base = {}
    for t in tens:   # (needs HP_NO_ARENA=1 to see every tensor)
        try:
            base[t] = e.debug_tensor(t, n)
        except Exception:
            pass
    bases.append(base)
hits = 0
REPS = int(os.environ.get("REPS", "60"))
for rep in range(REPS):
    for k, e in enumerate(engs):
        if k and delay_us:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e6 < delay_us:
                pass
        e.enqueue_u8(dev, n)
    for e in engs:
        e.synchronize()
    for k, e in enumerate(engs):
        r = outs(e)
        bad = [(i, sorted(set(np.argwhere(a != b)[:, 0]))) for i, (a, b) in enumerate(zip(ones[k], r)) if not np.array_equal(a, b)]
        if bad:
            hits += 1
            if hits <= 4:
                print("rep", rep, "engine", k, f"({dtypes[k]})", "outputs differ (output, frames):", bad)
                for t in sorted(bases[k]):
                    cur = e.debug_tensor(t, n)
                    if not np.array_equal(cur, bases[k][t]):
                        d = np.argwhere(cur != bases[k][t])
                        prod = [(L.op, L.cin, L.cout, L.kh, L.stride) for L in m.layers if L.out == t]
                        print("    first differing tensor", t, prod, "elements", len(d), "frames", sorted(set(d[:, 0])), "channels", d[:, 1].min(), "..", d[:, 1].max(), "max abs diff", float(np.abs(cur - bases[k][t]).max()))
                        break
import ctypes as C
cnt = (C.c_uint * 4)()
_lib.lib().hp_debug_first_conv_verify(cnt, 0)
print("first_conv32 LDS verify (HP_FIRST_CONV_VERIFY=1): patch words differing", cnt[0], "weight words differing", cnt[1], "blocks checked", cnt[2], "outputs whose second evaluation differed", cnt[3])
print(dtypes, "delay", delay_us, "us: mismatching (rep, engine) pairs:", hits, "of", 2 * REPS)
