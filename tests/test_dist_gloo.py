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
