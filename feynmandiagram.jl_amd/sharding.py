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


def make_comm(rank: int, world: int, group=None):
    """The C-ABI communicator (``fdg_comm_*``: RCCL bound inside libfdg.so) for callers that want the
    reduction without torch in the data path.  The 128-byte id travels over the already initialised
    ``torch.distributed`` group (any backend); a Julia or C host would use MPI or a file instead."""
    import torch.distributed as dist
    from . import capi
    box = [capi.Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return capi.Comm(box[0], rank, world)


def reduce_observable(acc, group=None, comm=None):
    """In-place sum of the per-rank accumulators over all ranks (no-op when
    torch.distributed is not initialised).  With ``comm`` (see make_comm) the sum
    goes through ``fdg_reduce_device`` on the current stream instead."""
    import torch.distributed as dist
    if comm is not None:
        import torch
        comm.reduce(acc.data_ptr(), acc.numel(), -1, torch.cuda.current_stream(acc.device).cuda_stream)
        return acc
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc
