"""GPU dev tool: does the evaluation rate still depend on where the roots land when they land NEXT TO the leaves?
Tile-major batch with separate leaf and root arrays (Array{Float64,3}(64, L, T) and (64, R, T)) against ONE array of
tile records (64, L + R, T) whose last R columns of every tile are the roots -- the same entry point
(fdg_eval_device_tiled: leaf tile stride = root tile stride = 64 (L + R), root base = leaf base + 512 L bytes), the same
algorithmic bytes.  Several allocations each, a pad of another size first (moves the allocator's state).
usage: gpu_tile_record_probe.py [workload] [B] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
shift_mb = [0, 517, 3, 2051, 64, 9000, 130, 1]


def timed(fn, n=8, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


frac = lambda ms: 8 * (L + R) * B / ms / 1e6 / 8000
for r in range(rounds):
    pad = torch.empty(max(1, shift_mb[r % len(shift_mb)]) << 20, dtype=torch.uint8, device=dev)
    # (a) separate arrays
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
    root = torch.empty((T, R, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    a = timed(lambda: h.eval_device_tiled(leaf.data_ptr(), 1, 64, 64 * L, root.data_ptr(), 1, 64, 64 * R, B, st))
    ref = root[:4096].clone()
    del leaf, root
    torch.cuda.empty_cache()
    # (b) tile records
    rec = torch.empty((T, L + R, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(rec.data_ptr(), B, L, 1, 64, 64 * (L + R), 1234, 0, st)
    rp = rec.data_ptr() + 512 * L
    b = timed(lambda: h.eval_device_tiled(rec.data_ptr(), 1, 64, 64 * (L + R), rp, 1, 64, 64 * (L + R), B, st))
    same = bool(torch.equal(rec[:4096, L:, :], ref))
    print(f"round {r} pad {shift_mb[r % len(shift_mb)]:5d} MB  separate {a:.3f} ms frac {frac(a):.3f}  |  tile records {b:.3f} ms frac {frac(b):.3f}  bits equal {same}", flush=True)
    del rec, pad
    torch.cuda.empty_cache()
