"""Generate the committed known-answer vectors for the PAF parser (the reference ships none).

    python tests/golden/make_golden.py

Inputs are the seeded synthetic heat-maps of hyperpose_amd/synth.py at a REDUCED size (so the fixture stays
small); outputs come from the strict-IEEE oracle build (oracle/_build/liboracle.so).  The GPU parity tests
compare libhp_hip.so against these files bit for bit, and the CPU suite re-checks the oracle against them
(guards the oracle itself against accidental edits).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperpose_amd import synth  # noqa: E402
from oracle import loader  # noqa: E402

CASES = [  # (rows, cols, people, seed-salt)
    (46, 54, 3, 1),   # config A geometry (non-square: anisotropic x4.696 / x3.407 up-sampling)
    (46, 46, 5, 2),   # square: exact x4 replication path
    (23, 31, 2, 3),   # small odd geometry
    (46, 54, 0, 4),   # empty frame: noise only
]


def main():
    out = {}
    meta = []
    for i, (rows, cols, people, salt) in enumerate(CASES):
        rng = synth.rng_for(1, salt=100 + salt)
        conf, paf, _ = synth.paf_maps(rng, 1, rows, cols, people=(people,),
                                      **({"scale_range": (8.0, 16.0)} if rows < 40 else {}))
        humans, peaks, conns = loader.paf_process(conf[0], paf[0])
        out[f"conf_{i}"] = conf[0].astype(np.float16).astype(np.float32)  # exactly representable -> compresses well
        out[f"paf_{i}"] = paf[0].astype(np.float16).astype(np.float32)
        humans, peaks, conns = loader.paf_process(out[f"conf_{i}"], out[f"paf_{i}"])
        out[f"humans_{i}"] = humans
        out[f"peaks_{i}"] = peaks
        out[f"conns_{i}"] = conns
        meta.append({"rows": rows, "cols": cols, "people": people, "n_humans": int(len(humans)),
                     "n_peaks": int(len(peaks)), "n_conns": int(len(conns))})
        print(meta[-1])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "paf_golden.npz"), **out)


if __name__ == "__main__":
    main()
