"""Multi-GPU plumbing of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

The path shards by independent units (images): rank r owns images [r*B/n, (r+1)*B/n).  The only
collective is ONE broadcast of the canonical weight blob at load time; steady state has none
(SURVEY.md §8e).  The reference has no multi-device code at all (one device, netFPGA.cpp:376).
"""
from __future__ import annotations

import os


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend, rank, world, local_rank=0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)   # gloo rehearsal of the GPU path: tensors still live on the rank's device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return torch, dist


def shard_bounds(global_batch, world, rank):
    """Contiguous image range of `rank`; the first (global_batch % world) ranks get one extra."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_blob(dist, blob_tensor, src=0):
    """Broadcast the weight blob (a uint8 tensor on the rank's device) from `src` in place."""
    dist.broadcast(blob_tensor, src=src)
    return blob_tensor


def max_over_ranks(torch, dist, value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(torch, dist, local_rows, world):
    """All-gather equally-shaped [rows, cols] tensors into one [world*rows, cols] tensor (used by tests
    and by callers that want the whole batch's logits on every rank)."""
    parts = [torch.empty_like(local_rows) for _ in range(world)]
    dist.all_gather(parts, local_rows)
    return torch.cat(parts, dim=0)
