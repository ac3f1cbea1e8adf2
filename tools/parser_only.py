"""parser-only loop (1 pipe) for rocprofv3: isolated durations of the PAF parser kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperpose_amd import _lib, synth
from hyperpose_amd.parser import Paf
_lib.init(0)
cfg = bench.config(1, "f16")   # BASELINE configs[1]: batch 8 @ 368 x 432 -> 46 x 54 maps
BATCH = cfg["batch"]
rng = synth.rng_for(1, salt=0)
conf, paf, _ = synth.paf_maps(rng, BATCH, cfg["h"] // 8, cfg["w"] // 8, people=cfg["people"])
cd, pd = _lib.DevBuf.from_numpy(conf), _lib.DevBuf.from_numpy(paf)
p = Paf(max_batch=BATCH)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    p.enqueue(cd, pd, BATCH, conf.shape[1:], paf.shape[1:])
    h = p.collect()
print(sum(len(x) for x in h), "humans in the last batch")
