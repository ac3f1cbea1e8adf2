"""PifPaf parser alone on synthetic maps (BASELINE configs[4] geometry, 64 frames, 1-6 people per frame): ms per batch through
hp_pifpaf_enqueue / hp_pifpaf_collect, the number of humans, and how many frames the device decoder handed to the host tail.
Run on the GPU box, optionally under `rocprofv3 --kernel-trace --stats` (see profiles/r02_pifpaf_parser_kernel_stats.csv):

    PYTHONPATH=. python tools/pifpaf_parser_bench.py            # device decoder (default)
    HP_PIFPAF_HOST_TAIL=1 PYTHONPATH=. python tools/pifpaf_parser_bench.py
"""
import time

import numpy as np

from hyperpose_amd import _lib as hp
from hyperpose_amd import synth
from hyperpose_amd.parser import PifPaf

B = 64
paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=5), B, people=(3, 4, 2, 5, 6, 1))
p = PifPaf(385, 385, max_batch=B)
dp, di = hp.DevBuf.from_numpy(paf), hp.DevBuf.from_numpy(pif)
ts = []
for it in range(22):
    t0 = time.perf_counter()
    p.enqueue(dp, di, B, 49, 49)
    got = p.collect()
    ts.append((time.perf_counter() - t0) * 1e3)
flags = p.decode_flags(B)
print({"ms_per_batch_median": round(float(np.median(ts[2:])), 3), "frames": B, "humans": sum(len(g) for g in got),
       "decoded_on_device": sum(f == 0 for f in flags), "host_tail": sum(f != 0 for f in flags)})
