"""GPU dev tool: does the headline's two-mode placement effect depend on the tile stride?  Tile-major batches whose tile stride is
padded by p doubles (the kernels take the tile strides as arguments), several allocations, evaluation and fused accumulation.
usage: gpu_tile_pad_probe.py [workload] [B] [rounds] [pads, comma separated, in doubles] [root pads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pads = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,16,32,64,96,128,256,512").split(",")]
rpads = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
pmax, rmax = max(pads), max(rpads)
shift_mb = [0, 517, 3, 2051, 64, 9000, 130, 1]
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros(R, dtype=torch.float64, device=dev)


def timed(fn, n=8, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(n))
    return d[0], d[len(d) // 2]


for r in range(rounds):
    pad = torch.empty(max(1, shift_mb[r % len(shift_mb)]) << 20, dtype=torch.uint8, device=dev)
    leaf = torch.empty(T * (64 * L + pmax), dtype=torch.float64, device=dev)
    root = torch.empty(T * (64 * R + rmax), dtype=torch.float64, device=dev)
    print(f"round {r}: leaf @ {leaf.data_ptr():#x} root @ {root.data_ptr():#x}", flush=True)
    for p in pads:
        for q in rpads:
            lt, rt = 64 * L + p, 64 * R + q
            capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, lt, 1234, 0, st)
            ev = lambda: h.eval_device_tiled(leaf.data_ptr(), 1, 64, lt, root.data_ptr(), 1, 64, rt, B, st)
            ac = lambda: h.accumulate_device_tiled(leaf.data_ptr(), 1, 64, lt, 0, acc.data_ptr(), B, st)
            (e, em), (a, am) = timed(ev), timed(ac)
            k = h.kernel_info()["last_kernel"]
            print(f"  pad {p:4d} rpad {q:4d}: eval min {e:.3f} med {em:.3f} ms frac {8 * (L + R) * B / em / 1e6 / 8000:.3f}   acc {am:.3f} ms frac {8 * L * B / am / 1e6 / 8000:.3f}  {k}", flush=True)
    del leaf, root, pad
    torch.cuda.empty_cache()
