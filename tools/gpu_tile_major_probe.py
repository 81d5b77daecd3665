"""GPU dev tool: the headline launch on a leaf-major matrix and on a tile-major batch (fdg_eval_device_tiled), side by side,
over several allocations in one process.  Question: does the tile-major layout -- one contiguous block of
512 L bytes per wave instead of L column streams B * 8 bytes apart -- remove the two placement modes of DESIGN.md 6a?
Each round: a pad of another size first (moves the allocation), both batches allocated and filled with the same Philox values,
roots compared bit for bit between the layouts, then `n` timed launches each (HIP events on the launch stream; min and mean).
usage: gpu_tile_major_probe.py [workload] [B] [rounds] [acc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
with_acc = len(sys.argv) > 4 and sys.argv[4] == "acc"
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
shift_mb = [0, 3, 517, 1, 2051, 64, 9000, 130, 33, 1027, 260, 4100]
st = torch.cuda.current_stream().cuda_stream
bytes_launch = 8 * (L + R) * B


def timed(fn, n=10, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
    return min(ms), sum(ms) / n


for r in range(rounds):
    pad = torch.empty(max(1, shift_mb[r % len(shift_mb)]) << 20, dtype=torch.uint8, device=dev)
    leaf_lm = torch.empty((L, B), dtype=torch.float64, device=dev)          # leaf-major: column stride B
    root_lm = torch.empty((R, B), dtype=torch.float64, device=dev)
    leaf_tm = torch.empty((T, L, 64), dtype=torch.float64, device=dev)      # tile-major: (tile, leaf, sample) strides (64 L, 64, 1)
    root_tm = torch.empty((T, R, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(leaf_lm.data_ptr(), B, L, 1, B, 1234, 0, st)
    capi.fill_uniform_device_tiled(leaf_tm.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    run_lm = lambda: h.eval_device(leaf_lm.data_ptr(), 1, B, root_lm.data_ptr(), 1, B, B, st)
    run_tm = lambda: h.eval_device_tiled(leaf_tm.data_ptr(), 1, 64, 64 * L, root_tm.data_ptr(), 1, 64, 64 * R, B, st)
    run_lm(); k_lm = f.kernel_info()["last_kernel"]
    run_tm(); k_tm = f.kernel_info()["last_kernel"]
    torch.cuda.synchronize()
    same = True
    for k in range(R):      # (per root: no second copy of the whole root array)
        a = root_tm[:, k, :].reshape(-1)[:B]
        same = same and bool(torch.equal(a.view(torch.int64), root_lm[k].view(torch.int64)))
    lm = timed(run_lm); tm = timed(run_tm)
    line = (f"round {r}: pad {shift_mb[r % len(shift_mb)]:5d} MB  bitwise equal {same}  "
            f"leaf-major ({k_lm}) min {lm[0]:.3f} mean {lm[1]:.3f} ms frac {bytes_launch / lm[0] / 1e6 / 8000:.3f}  |  "
            f"tile-major ({k_tm}) min {tm[0]:.3f} mean {tm[1]:.3f} ms frac {bytes_launch / tm[0] / 1e6 / 8000:.3f}")
    if with_acc:
        acc = torch.zeros(R, dtype=torch.float64, device=dev)
        a_lm = timed(lambda: h.accumulate_device(leaf_lm.data_ptr(), 1, B, 0, acc.data_ptr(), B, st))
        a_tm = timed(lambda: h.accumulate_device_tiled(leaf_tm.data_ptr(), 1, 64, 64 * L, 0, acc.data_ptr(), B, st))
        line += f"  ||  accumulate: leaf-major {8 * L * B / a_lm[0] / 1e6 / 8000:.3f}  tile-major {8 * L * B / a_tm[0] / 1e6 / 8000:.3f}"
    print(line, flush=True)
    del leaf_lm, root_lm, leaf_tm, root_tm, pad
    torch.cuda.empty_cache()
