"""Per-layer HIP-event timing table for a built-in topology (hp_engine_profile) — run on the GPU box.

    python tools/profile_layers.py [arch] [w] [h] [batch] [f16|f32]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperpose_amd import _lib  # noqa: E402
from hyperpose_amd.engine import Engine, Model  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "lw_openpose_mobilenet"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 432
h = int(sys.argv[3]) if len(sys.argv) > 3 else 368
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dtype = sys.argv[5] if len(sys.argv) > 5 else "f16"
_lib.init(0)
m = Model(arch, w, h)
eng = Engine.from_model(m, m.init_weights(1), max_batch=n, dtype=dtype)
prof = eng.profile(n, iters=20)                   # each step 20x back to back (weights warm in L2)
seq = eng.profile(n, iters=20, in_sequence=True)  # the schedule in order, events in between (what an inference sees)
eng2 = Engine.from_model(m, m.init_weights(1), max_batch=n, dtype=dtype)
pair = eng.profile(n, iters=20, pair=eng2)        # two instances on two streams: machine time per launch with overlapping pipes
names = {100: "sep", 101: "head", 102: "chain", 103: "bneck", 1: "conv", 2: "dw", 3: "pool", 4: "up"}
tot = 0.0
print(f"{'#':>3} {'op':5} {'cin':>4} {'cout':>4} k s d  {'tile':>8} {'us':>8} {'TF/s':>7} {'GB/s':>7} {'in-seq us':>9} {'pair us':>8}")
tot_seq = tot_pair = 0.0
for p, q, r in zip(prof, seq, pair):
    L = m.layers[p["layer"]]
    us = p["ms"] * 1e3
    tot += us
    print(f"{p['layer']:3d} {names[p['op']]:5} {L.cin:4d} {L.cout:4d} {L.kh} {L.stride} {L.dil}  {p['tile']:8d} {us:8.1f} "
          f"{p['flops'] / (us * 1e-6) / 1e12:7.1f} {p['bytes'] / (us * 1e-6) / 1e9:7.0f} {q['ms'] * 1e3:9.1f} {r['ms'] * 1e3:8.1f}")
    tot_seq += q["ms"] * 1e3
    tot_pair += r["ms"] * 1e3
print(f"total {tot:.1f} us per batch of {n} -> {n / (tot * 1e-6):.0f} FPS serial; {m.flops_per_frame * n / (tot * 1e-6) / 1e12:.1f} TF/s")
print(f"in sequence {tot_seq:.1f} us per batch of {n} -> {n / (tot_seq * 1e-6):.0f} FPS")
print(f"two streams {tot_pair:.1f} us per batch of {n} -> {n / (tot_pair * 1e-6):.0f} FPS")
