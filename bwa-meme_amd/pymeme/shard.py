"""Multi-GPU sharding helpers: reads are independent units and the index is read-only, so the path shards
with no data-path collective.  One process per GPU; the only collectives are the one-off
index broadcast at start-up and a result gather when a single host wants the whole chunk back in order."""
from __future__ import annotations

import torch
import torch.distributed as dist

BATCH = 512  # the reference's kt_for work unit (BATCH_SIZE, src/macro.h:48)


def partition(n_reads: int, world: int, batch: int = BATCH):
    """Contiguous, batch-aligned split of [0, n_reads) into `world` ranges (balanced to one batch).
    Returns [(lo, hi)] * world; concatenating the ranges in rank order restores the input order."""
    n_batches = (n_reads + batch - 1) // batch
    out, b0 = [], 0
    for r in range(world):
        nb = n_batches // world + (1 if r < n_batches % world else 0)
        lo, hi = min(b0 * batch, n_reads), min((b0 + nb) * batch, n_reads)
        out.append((lo, hi))
        b0 += nb
    return out


def broadcast_index(tensors, src: int = 0):
    """One-off broadcast of the index image (RCCL over xGMI with backend 'nccl', gloo in the CPU tests)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src)
    return tensors


def gather_in_order(local_obj, dst: int = 0):
    """Collect per-rank results on `dst` in rank order (host-side, after the timed region)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local_obj, out, dst=dst)
    return out


def max_over_ranks(seconds: float, device=None) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
