"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py (weight broadcast, frame sharding, max-over-ranks)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hyperpose_amd import dist as hd
    from hyperpose_amd.engine import Model
    d = hd.init("gloo")
    m = Model("lw_openpose_mobilenet", 96, 80)
    blob = m.init_weights(11) if rank == 0 else None
    w = hd.broadcast_weights(blob, m.n_weights, rank, world)
    start, cnt = hd.shard(13, rank, world)
    t = hd.max_over_ranks(1.0 + rank, world)
    s = hd.sum_over_ranks(cnt, world)
    q.put((rank, float(np.float64(w.astype(np.float64).sum())), int(w.size), start, cnt, t, s))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, sum0, n0, s0, c0, t0, tot0), (r1, sum1, n1, s1, c1, t1, tot1) = res
    assert sum0 == sum1 and n0 == n1 > 4_000_000     # identical weights on both ranks
    assert (s0, c0) == (0, 7) and (s1, c1) == (7, 6)   # contiguous, complete, disjoint
    assert t0 == t1 == 2.0 and tot0 == tot1 == 13.0


def _fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from hyperpose_amd import dist as hd
    backend = hd.init_for_gpu(torch.device("cuda", rank))  # no GPU here: RCCL cannot come up, the job must carry on over gloo
    dev = hd.collective_device(torch.device("cuda", rank))
    w = hd.broadcast_weights(np.arange(1000, dtype=np.float32) if rank == 0 else None, 1000, rank, world, device=dev)
    t = hd.max_over_ranks(3.0 + rank, world, device=dev)
    q.put((rank, backend, str(dev), float(w.sum()), t))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_failure_falls_back_to_gloo():
    """bench.py under torchrun on a node where the GPU backend cannot be initialised: the collectives (start-up broadcast, timing
    reductions) move to gloo on the host instead of killing the run."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, backend, dev, wsum, t in res:
        assert backend == "gloo" and dev == "cpu"
        assert wsum == float(np.arange(1000, dtype=np.float32).sum()) and t == 4.0


def _asymmetric_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 0:
        os.environ["HP_DIST_BACKEND"] = "gloo"  # set on ONE rank only
    import time

    import torch
    import torch.distributed as dist

    from hyperpose_amd import dist as hd
    if rank == 1:
        hd._rccl_usable = lambda: True  # this rank believes RCCL is fine and would go on to create the group
    t0 = time.perf_counter()
    backend = hd.init_for_gpu(torch.device("cuda", rank))
    q.put((rank, backend, time.perf_counter() - t0))
    dist.barrier()
    dist.destroy_process_group()


def test_backend_choice_is_agreed_before_any_rccl_call():
    """with HP_DIST_BACKEND=gloo on a subset of the ranks the others must not walk into `new_group(backend="nccl")` alone (they
    would sit in its store barrier until the timeout): the wish is reduced over gloo FIRST and every rank stays on gloo at once."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_asymmetric_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, backend, dt in res:
        assert backend == "gloo" and dt < 30.0, (rank, backend, dt)


def test_shard_covers_everything():
    from hyperpose_amd import dist as hd
    for total in (0, 1, 8, 13, 64):
        for world in (1, 2, 4, 8):
            spans = [hd.shard(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from hyperpose_amd import dist as hd
    d = hd.init("gloo")
    out = {}
    for k, cfg in bench.CONFIGS.items():
        for scaling in ("weak", "strong"):
            mine, glob = bench.rank_plan(cfg["batch"], scaling, rank, world)
            # what bench.measure reports: value = global frames per step * steps / MAX-over-ranks time
            total = hd.sum_over_ranks(mine, world)
            out[(k, scaling)] = (mine, glob, total)
    q.put((rank, out))
    d.barrier()
    d.destroy_process_group()


def test_bench_rank_plan_world2():
    """bench.py's per-rank work split: weak = every rank its own full batch (global = batch * world), strong = the
    configuration's batch sharded contiguously; the per-rank counts add up to the global batch the JSON line reports."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import bench
    from hyperpose_amd import dist as hd
    for k, cfg in bench.CONFIGS.items():
        b = cfg["batch"]
        for r in (0, 1):
            assert res[r][(k, "weak")] == (b, 2 * b, 2.0 * b)
            mine, glob, total = res[r][(k, "strong")]
            assert glob == b and total == float(b) and mine == hd.shard(b, r, 2)[1]   # (configs[0]: one frame, rank 1 idles)
    assert res[0][(3, "strong")][0] == 16 and res[0][(4, "strong")][0] == 32


def test_bench_gpus_flag_is_honoured(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (or fail loudly), never run one rank and
    print n_gpus = 1: on this GPU-less box the device check refuses with exit code 2."""
    import bench
    from hyperpose_amd import _lib
    assert bench.parse_args(["--gpus", "8"]).gpus == 8
    if _lib.lib().hp_device_count() >= 2:
        return  # a multi-GPU box would really launch; covered by the driver's SCALE runs
    assert bench.respawn(2) == 2
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        bench.main()
        raise AssertionError("bench.main() returned instead of re-launching / failing")
    except SystemExit as e:
        assert e.code == 2


def test_c_abi_shard_matches_python():
    """hp_dist_shard (the C ABI's frame split for C++ hosts) == hyperpose_amd.dist.shard."""
    import ctypes as C
    from hyperpose_amd import _lib
    from hyperpose_amd import dist as hd
    L = _lib.lib()
    for total in (0, 1, 8, 13, 32, 64):
        for world in (1, 2, 3, 8):
            for r in range(world):
                s, c = C.c_int(-1), C.c_int(-1)
                L.hp_dist_shard(total, r, world, C.byref(s), C.byref(c))
                assert (s.value, c.value) == hd.shard(total, r, world)
