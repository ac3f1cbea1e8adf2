"""Generate the committed known-answer vectors for the PAF, PoseProposal and PifPaf parsers FROM THE REFERENCE'S OWN
CODE (oracle/_ref/libhp_ref.so = src/paf.cpp + src/post_process.hpp, src/pose_proposal.cpp, src/pifpaf.cpp,
src/pifpaf_decoder/*.cpp compiled where they lie; only possible in a container that mounts /root/reference).
For PAF the reference's two OpenCV calls are the restatements of oracle/paf_oracle.cpp (OpenCV is not in the image);
everything else in the file - peaks, line integrals, std::sort + greedy assignment, assembly - is reference code.

    python tests/golden/make_golden_ref.py

Inputs are the seeded synthetic tensors of hyperpose_amd/synth.py stored as fp16-representable fp32 (small
files); outputs are the reference's human_t lists.  The GPU parity tests compare libhp_hip.so with these files
bit for bit, so they also run on the GPU box where /root/reference does not exist.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperpose_amd import synth  # noqa: E402
from oracle import loader  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def q16(a):
    return a.astype(np.float16).astype(np.float32)


PAF_CASES = [  # (rows, cols, people, seed-salt)
    (46, 54, 3, 1),   # config A geometry (non-square: anisotropic x4.696 / x3.407 up-sampling)
    (46, 46, 5, 2),   # square: exact x4 replication path
    (23, 31, 2, 3),   # small odd geometry
    (46, 54, 0, 4),   # empty frame: noise only
]


def paf():
    out, meta = {}, []
    for i, (rows, cols, people, salt) in enumerate(PAF_CASES):
        rng = synth.rng_for(1, salt=100 + salt)
        conf, pafm, _ = synth.paf_maps(rng, 1, rows, cols, people=(people,),
                                       **({"scale_range": (8.0, 16.0)} if rows < 40 else {}))
        out[f"conf_{i}"], out[f"paf_{i}"] = q16(conf[0]), q16(pafm[0])  # exactly representable -> compresses well
        humans, peaks, conns = loader.ref_paf_process(out[f"conf_{i}"], out[f"paf_{i}"])
        out[f"humans_{i}"], out[f"peaks_{i}"], out[f"conns_{i}"] = humans, peaks, conns
        meta.append({"rows": rows, "cols": cols, "people": people, "n_humans": int(len(humans)),
                     "n_peaks": int(len(peaks)), "n_conns": int(len(conns))})
        print("paf", meta[-1])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "paf_golden.npz"), **out)


def ppn():
    out, meta = {}, []
    cases = [(1, 31), (3, 32), (5, 33), (0, 34), (8, 35)]
    for i, (people, salt) in enumerate(cases):
        t = synth.ppn_maps(synth.rng_for(3, salt=salt), 1, people=(people,), spurious=0.02 if i % 2 else 0.005)
        t = [q16(a[0]) for a in t]
        humans = loader.ref_ppn_process(t)
        for k, a in enumerate(t):
            out[f"t{k}_{i}"] = a.astype(np.float16)  # exactly representable: tests widen back to fp32
        out[f"humans_{i}"] = humans
        meta.append({"people": people, "n_humans": int(len(humans))})
        print("ppn", meta[-1])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "ppn_golden.npz"), **out)


def pifpaf():
    out, meta = {}, []
    cases = [(1, 49, 49, 41), (3, 49, 49, 42), (6, 49, 49, 43), (0, 49, 49, 44), (2, 33, 41, 45)]
    for i, (people, fh, fw, salt) in enumerate(cases):
        paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=salt), 1, fh, fw, people=(people,))
        paf, pif = q16(paf[0]), q16(pif[0])
        net_h, net_w = (fh - 1) * 8 + 1, (fw - 1) * 8 + 1
        humans = loader.ref_pifpaf_process(paf, pif, net_h, net_w)
        out[f"paf_{i}"], out[f"pif_{i}"] = paf.astype(np.float16), pif.astype(np.float16)
        out[f"humans_{i}"] = humans
        meta.append({"people": people, "fh": fh, "fw": fw, "net_h": net_h, "net_w": net_w, "n_humans": int(len(humans))})
        print("pifpaf", meta[-1])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "pifpaf_golden.npz"), **out)


if __name__ == "__main__":
    assert loader.ref_lib() is not None, "oracle/_ref not built (needs /root/reference)"
    assert loader.have_ref_paf(), "oracle/_ref is older than the PAF entry points: make -C oracle ref"
    paf()
    ppn()
    pifpaf()
