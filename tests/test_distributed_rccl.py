"""N > 1 on real devices: one process per GPU (world = 2, 4, 8 -- whichever the node has), the C-ABI communicator (fdg_comm_unique_id / fdg_comm_create /
fdg_reduce_device: RCCL inside libfdg.so, no torch.distributed anywhere) around sharded evaluation + fused accumulation
of the GV 5th-order self-energy (BASELINE.json config 5 in miniature).  Each world size is skipped when fewer GPUs are visible
(the 1-GPU box of `gpurun` skips all three); the decomposition itself is covered on CPU by test_distributed_gloo.py, the driver's
own N = 8 command line by tests/test_bench_line.py::test_dry_run_with_eight_ranks_is_config_5_as_baseline_words_it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, id_path, n_total, q):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import feynmandiagram_jl_amd as fd
    from feynmandiagram_jl_amd import capi, workloads
    from feynmandiagram_jl_amd.sharding import shard_range
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    # the 128-byte id travels through a file (a Julia host would use MPI.jl or a file in exactly this way)
    if rank == 0:
        ident = capi.Comm.unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(ident)
        os.rename(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 120:
                raise TimeoutError("no communicator id")
            time.sleep(0.05)
        ident = open(id_path, "rb").read()
    comm = capi.Comm(ident, rank, world)
    t = workloads.get("gv_sigma5")
    f = fd.compile_table(t, specialize="isa")
    start, count = shard_range(n_total, rank, world)
    leaf = torch.empty((t.n_leaf, count), dtype=torch.float64, device=dev).t()
    st = torch.cuda.current_stream().cuda_stream
    capi.fill_uniform_device(leaf.data_ptr(), count, t.n_leaf, leaf.stride(0), leaf.stride(1), 1234, start, st)
    acc = f.accumulate(leaf)                       # weight 1: acc[k] = sum over this rank's shard of root_k
    absroot = f(None, leaf).abs().sum(dim=0)
    comm.reduce(acc.data_ptr(), acc.numel(), -1, st)       # all-reduce: every rank holds the total
    comm.reduce(absroot.data_ptr(), absroot.numel(), 0, st)  # rooted form: rank 0 holds the total
    torch.cuda.synchronize()
    q.put((rank, acc.cpu().numpy(), absroot.cpu().numpy()))
    comm.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gpus_shard_and_reduce_through_the_c_abi(tmp_path, world):
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (the final observable reduce over xGMI)")
    import torch.multiprocessing as mp
    import oracle
    from feynmandiagram_jl_amd import workloads
    n_total = 40_001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, str(tmp_path / "comm_id"), n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, acc, absroot = q.get(timeout=600)
        res[r] = (acc, absroot)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    t = workloads.get("gv_sigma5")
    ref = oracle.eval_static(t, oracle.philox_uniform(n_total, t.n_leaf, 1234))
    scale = np.abs(ref).sum(axis=0)
    assert all(np.array_equal(res[0][0], res[r][0]) for r in range(1, world))         # every rank holds the same total
    assert np.all(np.abs(res[0][0] - ref.sum(axis=0)) <= 1e-12 * scale)             # sum order differs with the rank count
    assert np.allclose(res[0][1], scale, rtol=1e-12)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("comm", ["fdg", "torch"])
def test_the_drivers_own_multi_gpu_command_line(world, comm):
    """the first multi-GPU node that runs the suite also runs the driver's exact command -- `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` --
    with both carriers of the one collective (RCCL through libfdg's fdg_comm_*, and torch.distributed's nccl backend), and reads the
    one stdout line: whole-job value, n_gpus, weak scaling, config 5's shards.  No scaling curve exists until this has run on such a node."""
    import json
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--comm", comm, "--samples", "4000000", "--no-mc-step",
           "--secondary", "sigma2:tile_major"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["steps"] == 2 and line["value"] > 0
    c5 = line["config5"]
    assert "error" not in c5 and c5["n_gpus"] == world and c5["total_samples"] >= 1_000_000_000
