"""Soak test of the PifPaf device decoder against the host tail on random synthetic maps (run on the GPU box):
    PYTHONPATH=. python tools/pifpaf_stress.py [batches] [max people per frame, default 24] [noise amplitude: default a random one of 0.02 .. 0.28]
Prints the number of frames compared, how many the device decoder handed back to the host tail (and why), the largest number of humans a frame
returned, and any mismatch.  `4 300 0.6`: crowds far beyond anything real - several hundred annotations per frame (round 6: PD_MAXA / PD_Q 256 -> 1024)."""
import os
import sys

import numpy as np

from hyperpose_amd import synth
from hyperpose_amd.parser import PifPaf

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
max_people = int(sys.argv[2]) if len(sys.argv) > 2 else 24
noise_arg = float(sys.argv[3]) if len(sys.argv) > 3 else None
B = 32
os.environ["HP_PIFPAF_HOST_TAIL"] = "0"
dev = PifPaf(385, 385, 0.05, max_batch=B, cap_per_frame=1024)
os.environ["HP_PIFPAF_HOST_TAIL"] = "1"
host = PifPaf(385, 385, 0.05, max_batch=B, cap_per_frame=1024)
rng = np.random.default_rng(20244)
frames = fell = humans = bad = most = 0
why = {}
for it in range(n_batches):
    people = tuple(int(v) for v in rng.integers(max_people // 2 if max_people > 24 else 0, max_people + 1, 8))
    noise = noise_arg if noise_arg is not None else float(rng.choice([0.02, 0.1, 0.2, 0.28]))
    paf, pif = synth.pifpaf_maps(np.random.default_rng(1000 + it), B, people=people, noise=noise)
    a, b = dev.process_batch(paf, pif), host.process_batch(paf, pif)
    fl = dev.decode_flags(B)
    for f in range(B):
        frames += 1
        fell += fl[f] != 0
        if fl[f]:
            why[int(fl[f])] = why.get(int(fl[f]), 0) + 1
        humans += len(b[f])
        most = max(most, len(b[f]))
        if a[f].tobytes() != b[f].tobytes():
            bad += 1
            print("MISMATCH batch", it, "frame", f, "flags", fl[f], len(a[f]), len(b[f]))
print({"frames": frames, "humans": humans, "most_humans_in_a_frame": most, "host_tail_frames": int(fell), "decline_flags": why, "mismatches": bad})
