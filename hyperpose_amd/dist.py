"""Multi-GPU plumbing of the hot path: one process per GPU, frames sharded, ONE collective at start-up.

The reference has no multi-GPU inference path at all.  Frames are independent
(src/paf.cpp:338-339: parser scratch is overwritten per call; the network has no cross-frame state), so the path
shards with no steady-state exchange: every rank runs its own engine + parser on its own frames.  The only
collective is the broadcast of the weight blob from rank 0 (RCCL over xGMI on the GPU box, gloo in the CPU
tests) so that all ranks run the same network.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


_COLLECTIVE_DEVICE = None  # device the data collectives run on: the GPU under RCCL, "cpu" on gloo
_GROUP = None              # process group of the data collectives (None = the default group)
_BACKEND = None
LAST_BROADCAST = {}        # {"backend", "bytes", "ms"} of the last broadcast_weights (reported by bench.py)


def init_for_gpu(device, probe: bool = True):
    """Bring up the job's collectives on a GPU node; returns the name of the backend that carries them ("nccl" = RCCL, or "gloo").

    The default group is gloo (host TCP: rendezvous, barriers - it always comes up); the start-up weight broadcast and the timing
    reductions run on an RCCL group created next to it.  Whether RCCL is usable is decided COLLECTIVELY: every rank tries to create
    the group and to all-reduce one element on it, the ranks then take the minimum of their success flags over gloo, and either all
    of them use RCCL or all of them stay on gloo - no rank can end up alone in the other backend, no second rendezvous, no second
    port.  `HP_DIST_BACKEND=gloo` (set for all ranks by the launcher) skips RCCL.  The hot path itself has no collective."""
    global _COLLECTIVE_DEVICE, _GROUP, _BACKEND
    import datetime
    import sys

    import torch
    import torch.distributed as dist
    init("gloo")
    ok, why, group = 1, "", None
    if os.environ.get("HP_DIST_BACKEND", "nccl") == "gloo":
        ok, why = 0, "HP_DIST_BACKEND=gloo"
    else:
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            if probe:
                t = torch.ones(1, device=device)
                dist.all_reduce(t, group=group)
                torch.cuda.synchronize(device)
                if int(t.item()) != dist.get_world_size():
                    raise RuntimeError(f"probe all-reduce returned {t.item()}")
        except Exception as e:  # noqa: BLE001 - any failure of the GPU backend takes the same way out
            ok, why = 0, f"{type(e).__name__}: {e}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # gloo: the agreement
    if int(flag.item()) == 1:
        _GROUP, _COLLECTIVE_DEVICE, _BACKEND = group, device, "nccl"
    else:
        if why:
            print(f"[hyperpose_amd.dist] rank {dist.get_rank()}: RCCL not used ({why}); all ranks carry the collectives over gloo",
                  file=sys.stderr, flush=True)
        _GROUP, _COLLECTIVE_DEVICE, _BACKEND = None, "cpu", "gloo"
    return _BACKEND


def backend_name():
    """Backend of the data collectives after init_for_gpu / init ("nccl", "gloo") or None before."""
    if _BACKEND is not None:
        return _BACKEND
    try:
        import torch.distributed as dist
        return dist.get_backend() if dist.is_initialized() else None
    except Exception:  # noqa: BLE001
        return None


def collective_device(default):
    return default if _COLLECTIVE_DEVICE is None else _COLLECTIVE_DEVICE


def broadcast_weights(blob: np.ndarray | None, n: int, rank: int, world: int, device="cpu") -> np.ndarray:
    """Rank 0 passes the fp32 blob, the others None; every rank returns the same `n` floats."""
    if world == 1:
        assert blob is not None
        return blob
    import time

    import torch
    import torch.distributed as dist
    if rank == 0:
        assert blob is not None and blob.size == n
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    on_gpu = str(device) != "cpu"
    if on_gpu:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    dist.broadcast(t, src=0, group=_GROUP)
    if on_gpu:
        torch.cuda.synchronize(device)
    LAST_BROADCAST.update({"backend": backend_name(), "bytes": int(n) * 4, "ms": round((time.perf_counter() - t0) * 1e3, 3)})
    return t.cpu().numpy()


def shard(total_frames: int, rank: int, world: int):
    """Contiguous split of a global batch (B/8 frames per GPU): returns (start, count)."""
    base, rem = divmod(total_frames, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def max_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_GROUP)
    return float(t.item())


def sum_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_GROUP)
    return float(t.item())
