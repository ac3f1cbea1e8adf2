"""tools/r6_count_ann.py - the lattice frames of tests/test_pifpaf_device_gpu.py::test_hundreds_of_annotations_stay_on_the_device: decode flags, humans, device == host tail
(built once with PD_MAXA = 256 to confirm that these frames exceed the old capacity: flags [1, 1, 1, 1])."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from hyperpose_amd import _lib
_lib.init(0)
import test_pifpaf_device_gpu as T
B = 4
from hyperpose_amd import synth
paf, pif = T._lattice_maps(B)
ppaf, ppif = synth.pifpaf_maps(synth.rng_for(4, salt=55), B, people=(5, 6, 4, 7), noise=0.0)
person = ppif[:, :, 0] > 0.05
for c in range(5):
    pif[:, :, c][person] = ppif[:, :, c][person]
paf = ppaf.astype(np.float32)
for thr in (0.05,):
    dev, host = T._parser(False, 385, 385, thr, max_batch=B, cap_per_frame=1024), T._parser(True, 385, 385, thr, max_batch=B, cap_per_frame=1024)
    got, ref = dev.process_batch(paf, pif), host.process_batch(paf, pif)
    print("thr", thr, "flags", dev.decode_flags(B), "humans", [len(g) for g in got], [len(r) for r in ref], "equal", [g.tobytes() == r.tobytes() for g, r in zip(got, ref)], "strong cells", [int((pif[b, :, 0] >= 0.5).sum()) for b in range(B)])
