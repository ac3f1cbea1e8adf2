import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from hyperpose_amd import _lib, synth
from hyperpose_amd.engine import Model
_lib.init(0)
model = Model(bench.ARCH, bench.IN_W, bench.IN_H)
w = model.init_weights(20241)
rng = synth.rng_for(1, salt=0)
frames = synth.images_u8(rng, bench.BATCH, bench.IN_H, bench.IN_W)
conf, paf, _ = synth.paf_maps(rng, bench.BATCH, bench.IN_H // 8, bench.IN_W // 8, people=(1, 2, 4, 8, 16, 3, 5, 6))
fd, cd, pd = _lib.DevBuf.from_numpy(frames), _lib.DevBuf.from_numpy(conf), _lib.DevBuf.from_numpy(paf)
p = bench.Pipe(model, w, cd, pd)
for _ in range(20):
    p.submit(fd, True); p.collect()
ts = tc = 0.0
N = 200
for _ in range(N):
    t0 = time.perf_counter(); p.submit(fd, True); t1 = time.perf_counter()
    p.eng.synchronize(); import ctypes
    time.sleep(0.002)
    t2 = time.perf_counter(); p.collect(); t3 = time.perf_counter()
    ts += t1 - t0; tc += t3 - t2
print(f"submit {ts/N*1e6:.1f} us, collect (already complete) {tc/N*1e6:.1f} us")
