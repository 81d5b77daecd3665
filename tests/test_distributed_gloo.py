"""N > 1 path on CPU: world_size-2 gloo.  Each rank evaluates its shard of the
samples (the oracle stands in for the GPU kernel here -- this test is about the
decomposition: shard ranges, global Philox counters, the single all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feynmandiagram_jl_amd.sharding import reduce_observable, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                s, c = shard_range(n, r, world)
                assert s == pos and c >= 0
                pos += c
            assert pos == n
            counts = [shard_range(n, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    from feynmandiagram_jl_amd import workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = workloads.get("sigma2")
        start, count = shard_range(n_total, rank, world)
        leaf = oracle.philox_uniform(count, t.n_leaf, 1234, sample_offset=start)
        root = oracle.eval_static(t, leaf)
        acc = torch.from_numpy(root.sum(axis=0))
        absacc = torch.from_numpy(np.abs(root).sum(axis=0))
        dist.barrier()
        reduce_observable(acc)
        reduce_observable(absacc)
        if rank == 0:
            q.put((acc.numpy().copy(), absacc.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_reduce_matches_single_process():
    import oracle
    from feynmandiagram_jl_amd import workloads
    n_total, world = 20001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    acc, absacc = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = workloads.get("sigma2")
    leaf = oracle.philox_uniform(n_total, t.n_leaf, 1234)
    ref = oracle.eval_static(t, leaf)
    # fp64 sum order differs with the number of ranks: 1e-12 * sum|x|
    assert np.all(np.abs(acc - ref.sum(axis=0)) <= 1e-12 * np.abs(ref).sum(axis=0))
    assert np.allclose(absacc, np.abs(ref).sum(axis=0), rtol=1e-12)
