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


def _rccl_usable() -> bool:
    """Whether THIS rank could take part in an RCCL group at all (a local property, checked before any collective call on it)."""
    if os.environ.get("HP_DIST_BACKEND", "nccl") == "gloo":
        return False
    try:
        import torch
        import torch.distributed as dist
        return bool(dist.is_nccl_available() and torch.cuda.is_available())
    except Exception:  # noqa: BLE001
        return False


def init_for_gpu(device, probe: bool = True, probe_timeout_s: float = 45.0):
    """Bring up the job's collectives on a GPU node; returns the name of the backend that carries them ("nccl" = RCCL, or "gloo").

    The default group is gloo (host TCP: rendezvous, barriers - it always comes up); the start-up weight broadcast and the timing
    reductions run on an RCCL group created next to it.  Whether RCCL is used is decided COLLECTIVELY, in two rounds over gloo, so that
    no rank can end up alone inside an RCCL call:
      1. before anything touches RCCL the ranks take the MINIMUM of "this rank wants and can load RCCL" (`HP_DIST_BACKEND=gloo` on any
         one rank, a torch build without NCCL, no visible GPU -> every rank stays on gloo and nobody calls `new_group`);
      2. then every rank creates the group and runs ONE probe all-reduce with a bounded wait (`probe_timeout_s`; the watchdog's
         process-abort is switched off for it, so a rank whose peers failed gets an exception instead of being killed at the group's
         timeout), and the ranks take the minimum of their success flags: either all use RCCL or all stay on gloo.
    The probe always runs (`probe` is kept for source compatibility): a group nobody has tried is never committed to.  The hot path
    itself has no collective."""
    global _COLLECTIVE_DEVICE, _GROUP, _BACKEND
    import datetime
    import sys

    import torch
    import torch.distributed as dist
    init("gloo")
    del probe
    want = torch.tensor([1 if _rccl_usable() else 0], dtype=torch.int32)
    dist.all_reduce(want, op=dist.ReduceOp.MIN)  # gloo: round 1
    ok, why, group = 1, "", None
    if int(want.item()) == 0:
        ok, why = 0, "a rank asked for gloo (HP_DIST_BACKEND) or cannot load RCCL"
    else:
        # a failed collective must surface as an exception on the ranks that wait for it, not as the watchdog aborting the process
        # (for the PROBE only: the variable is restored below, and the group that carries the data collectives - weight broadcast, timing
        # reductions - is created afterwards with torch's normal error handling and its default 10-minute timeout, so a peer that dies
        # later makes the others fail at that timeout instead of hanging in synchronize().  work.wait(timeout) blocking the CPU needs
        # torch >= 2.6; this image has 2.10.)
        prev = os.environ.get("TORCH_NCCL_ASYNC_ERROR_HANDLING")
        os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
        wait = datetime.timedelta(seconds=probe_timeout_s)
        try:
            group = dist.new_group(backend="nccl", timeout=wait)
            t = torch.ones(1, device=device)
            work = dist.all_reduce(t, group=group, async_op=True)
            if work.wait(wait) is False:
                raise RuntimeError("probe all-reduce did not complete")
            torch.cuda.synchronize(device)
            if int(t.item()) != dist.get_world_size():
                raise RuntimeError(f"probe all-reduce returned {t.item()}")
        except Exception as e:  # noqa: BLE001 - any failure of the GPU backend takes the same way out
            ok, why = 0, f"{type(e).__name__}: {e}"
        finally:
            if prev is None:
                os.environ.pop("TORCH_NCCL_ASYNC_ERROR_HANDLING", None)
            else:
                os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = prev
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # gloo: round 2, the agreement
    if int(flag.item()) == 1:
        # every rank got here with a working RCCL: the data collectives' own group (collective call, same order on all ranks).  Whether that group
        # exists is AGREED on like the probe was: if new_group failed on some ranks only, the ranks would hold different groups and
        # the next broadcast would hang until its time-out - so either all ranks use the new group or all keep the probe's.
        fresh, made = None, 1
        try:
            fresh = dist.new_group(backend="nccl")
        except Exception:  # noqa: BLE001
            made = 0
        agreed = torch.tensor([made], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)  # gloo (the default group)
        if int(agreed.item()) == 1:
            group = fresh
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
