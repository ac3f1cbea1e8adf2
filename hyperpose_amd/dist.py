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


_COLLECTIVE_DEVICE = None  # device the collectives run on: the GPU under RCCL, "cpu" after a fall-back to gloo


def init_for_gpu(device, probe: bool = True):
    """RCCL (`nccl`) for the start-up broadcast and the two timing reductions; if the backend cannot be brought up on this node
    (raises at init or at the first collective) the job falls back to gloo on the host - the hot path itself has no collective, so
    nothing measured changes.  Returns the backend name."""
    global _COLLECTIVE_DEVICE
    import sys

    import torch
    import torch.distributed as dist
    try:
        init("nccl", device=device)
        if probe:
            t = torch.ones(1, device=device)
            dist.all_reduce(t)
            torch.cuda.synchronize(device)
        _COLLECTIVE_DEVICE = device
        return "nccl"
    except Exception as e:  # noqa: BLE001 - any failure of the GPU backend takes the same way out
        print(f"[hyperpose_amd.dist] RCCL unavailable ({type(e).__name__}: {e}); collectives fall back to gloo", file=sys.stderr, flush=True)
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + 1)  # the old store may still hold the port
        init("gloo")
        _COLLECTIVE_DEVICE = "cpu"
        return "gloo"


def collective_device(default):
    return default if _COLLECTIVE_DEVICE is None else _COLLECTIVE_DEVICE


def broadcast_weights(blob: np.ndarray | None, n: int, rank: int, world: int, device="cpu") -> np.ndarray:
    """Rank 0 passes the fp32 blob, the others None; every rank returns the same `n` floats."""
    if world == 1:
        assert blob is not None
        return blob
    import torch
    import torch.distributed as dist
    if rank == 0:
        assert blob is not None and blob.size == n
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(t, src=0)
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
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
