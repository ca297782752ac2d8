"""Multi-GPU helpers: one process per GPU, independent puzzles per rank.

The path shards by puzzle: attention is within a puzzle and BatchNorm statistics
are per process in the reference (no SyncBN), so inference needs no exchange at all; the only
collectives there are the timing barrier / max-over-ranks clock of the benchmark and an optional final
metric reduction (mirrors `self.log(..., sync_dist=True)`, auto_aggl.py:366-369).
Training has exactly one exchange per step — the gradient all-reduce of the DenoiserTransformer parameters
(Lightning DDP in the reference) — done by GradExchange below over RCCL (backend "nccl" on ROCm).
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


def balanced_assignment(fragment_counts: List[int], world: int, equal_count: bool = False) -> List[List[int]]:
    """greedy longest-processing-time assignment of puzzles to ranks by valid-fragment count
    (encoder work per puzzle varies ~10x with the fragment count).
    equal_count: every rank gets the same number of puzzles (training: the per-rank batch size is fixed, and the sparse
    embedding-gradient exchange needs equal shapes) — the heaviest unassigned puzzle goes to the least-loaded rank that still
    has a free slot; len(fragment_counts) must then be a multiple of world."""
    n = len(fragment_counts)
    if equal_count and n % world:
        raise ValueError("balanced_assignment(equal_count=True): the number of puzzles must be a multiple of the world size")
    cap = n // world if equal_count else n
    order = sorted(range(n), key=lambda i: (-fragment_counts[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min((k for k in range(world) if len(out[k]) < cap), key=lambda k: (loads[k], k))
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


class GradExchange:
    """Data-parallel gradient exchange over a flat gradient buffer (pfpp_hip.train.FlatParams.grads).

    The backward finishes the layers last-to-first; layer_done(i) starts the all-reduce of that layer's
    contiguous slice at once (async: it overlaps the rest of the backward; on xGMI's point-to-point links a few
    large messages beat many small ones, so the unit is a whole layer, 38 MB), all_done() sends the two remaining
    slices (AdaLN tables / embeddings / heads, which only complete at the very end), finish() waits and returns
    the factor that turns the summed gradients into the mean (folded into the optimizer kernel, no extra pass).
    Works on any backend (tested with gloo on CPU tensors)."""

    def __init__(self, grads: torch.Tensor, layer_ranges: List[Tuple[int, int]], sparse_range: Tuple[int, int] = (0, 0)):
        self.grads = grads
        self.layer_ranges = list(layer_ranges)
        # [a, b) of the head slice that is NOT all-reduced: the timestep-embedding tables (a third of all parameters) get
        # gradients in only `batch` of their 3072 rows per step, so the ranks exchange those rows instead (gather_rows)
        self.sparse_range = sparse_range
        self._handles: List[object] = []
        self._early: List[Tuple[int, int]] = []      # ranges of the head slice already reduced with a layer (AdaLN linears)
        self.enabled = True          # False: the owner accumulates locally (DenoiserTrainEngine.no_sync), nothing is reduced

    @staticmethod
    def active() -> bool:
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def reducing(self) -> bool:
        return self.enabled and self.active()

    def _reduce(self, a: int, b: int) -> None:
        import torch.distributed as dist

        if b <= a:
            return
        self._handles.append(dist.all_reduce(self.grads[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def layer_done(self, i: int, extra=()) -> None:
        """reduce layer i's slice, plus `extra` [a, b) ranges of the head slice whose gradients are final with this layer"""
        if self.reducing():
            self._reduce(*self.layer_ranges[i])
            for a, b in extra:
                self._reduce(a, b)
                self._early.append((a, b))

    def all_done(self, dense: bool = False) -> None:
        """the slices outside the layers.  dense = True also reduces the sparse range (gradient accumulation: the table rows of
        several micro-batches were scatter-added locally, there is no single (rows, index) pair to exchange)"""
        if self.reducing():
            a, b = self.sparse_range
            first = self.layer_ranges[0][0]
            skip = sorted(self._early + ([(a, b)] if (b > a and not dense) else []))
            pos = 0
            for x, y in skip + [(first, first)]:          # the head slice minus what went with the layers / goes as rows
                self._reduce(pos, min(x, first))
                pos = max(pos, y)
            self._reduce(self.layer_ranges[-1][1], self.grads.numel())
        self._early = []

    def gather_rows(self, rows: torch.Tensor, index: torch.Tensor, dim: int = 1):
        """every rank's (rows, index) concatenated along `dim` / 0 — the sparse form of an embedding-gradient exchange: the
        caller scatter-adds ALL ranks' rows into its own table gradient, which then equals the all-reduced (summed) dense
        gradient.  Needs the same `rows` / `index` shapes on every rank (equal per-rank batch size)."""
        import torch.distributed as dist

        world = dist.get_world_size()
        rs = [torch.empty_like(rows) for _ in range(world)]
        ix = [torch.empty_like(index) for _ in range(world)]
        dist.all_gather(rs, rows.contiguous())
        dist.all_gather(ix, index.contiguous())
        return torch.cat(rs, dim).contiguous(), torch.cat(ix, 0).contiguous()

    def wait_pending(self) -> None:
        """the reductions issued so far happen before whatever the CURRENT stream is given next (RCCL: a stream-side wait, the
        host does not block; gloo: the host waits)"""
        for h in self._handles:
            h.wait()
        self._handles = []

    @staticmethod
    def mean_factor() -> float:
        import torch.distributed as dist

        return 1.0 / dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1.0

    def finish(self) -> float:
        import torch.distributed as dist

        """wait for the reductions; returns the factor that turns the summed gradients into the mean.  Call it (through
        DenoiserTrainEngine.finish_grad_exchange) before reading gradients for clipping / logging: until then they are summed,
        not averaged, and reductions may still be in flight."""
        for h in self._handles:
            h.wait()
        self._handles = []
        return 1.0 / dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1.0
