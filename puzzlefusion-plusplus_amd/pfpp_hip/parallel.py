"""Multi-GPU helpers: one process per GPU, independent puzzles per rank, no data-path collective.

The path shards by puzzle: attention is within a puzzle and BatchNorm statistics
are per process in the reference (no SyncBN), so inference needs no exchange at all; the only
collectives are the timing barrier / max-over-ranks clock of the benchmark and an optional final
metric reduction (mirrors `self.log(..., sync_dist=True)`, auto_aggl.py:366-369).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous split of `total` puzzles over `world` ranks; the first total % world ranks get one more"""
    if not (0 <= rank < world):
        raise ValueError("rank outside [0, world)")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def balanced_assignment(fragment_counts: List[int], world: int) -> List[List[int]]:
    """greedy longest-processing-time assignment of puzzles to ranks by valid-fragment count
    (encoder work per puzzle varies ~10x with the fragment count)"""
    order = sorted(range(len(fragment_counts)), key=lambda i: -fragment_counts[i])
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += fragment_counts[i]
    return out


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
