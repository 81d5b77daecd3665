"""Multi-GPU decomposition of the evaluator: samples are independent, so they
shard across ranks with no data-path collective; the only exchange is ONE
all-reduce of the accumulated observable (R doubles) at the end.
The reference has no counterpart (single-threaded); downstream users do this
reduce in MCIntegration.jl.

One process per GPU, ``torch.distributed`` (backend "nccl" == RCCL on ROCm over
xGMI; "gloo" for the CPU tests).  The message is R * 8 bytes: latency-bound, so
nothing is bucketed or overlapped.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous sample range ``[start, start + count)`` of ``rank``; the first
    ``n_total % world`` ranks get one sample more.  The Philox counter of a
    sample is its global index, so results do not depend on ``world``."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def reduce_observable(acc, group=None):
    """In-place sum of the per-rank accumulators over all ranks (no-op when
    torch.distributed is not initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc
