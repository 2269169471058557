"""Blob sharding across GPUs: blobs are independent, so the batch is cut into contiguous
ranges balanced by work (nface + nvert, known from the headers); one process per GPU decodes its range;
there is no data-path collective.  torch.distributed is used only for the timing barrier and the
max-over-ranks reduction that bench.py's contract asks for (backend nccl = RCCL on GPUs, gloo in CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def balanced_ranges(weights: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) per rank with prefix-sum cut points closest to k/world of the total weight."""
    n = len(weights)
    total = float(sum(weights))
    if world <= 0:
        raise ValueError("world must be positive")
    cuts, acc, k = [0], 0.0, 1
    for i, w in enumerate(weights):
        # put the cut before item i if that is closer to the ideal than after it
        while k < world and acc + w / 2.0 >= k * total / world:
            cuts.append(i)
            k += 1
        acc += w
    while len(cuts) < world:
        cuts.append(n)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def my_range(weights: Sequence[int], world: int, rank: int) -> Tuple[int, int]:
    return balanced_ranges(weights, world)[rank]


def barrier(dist=None, device_sync=None):
    """barrier + device sync on both sides of a timed region"""
    if device_sync is not None:
        device_sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if device_sync is not None:
        device_sync()


def max_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
