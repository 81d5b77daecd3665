"""GPU dev tool (round 5): does config 5's per-step batch size matter?  Fused accumulation and evaluation of gv_sigma5 (tile-major) at several
batch sizes, same process, waits for the driver's wipe between sizes.   usage: gpu_step_size.py [workload] [sizes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "gv_sigma5"
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2000000,4000000,8000000,16000000,2000000").split(",")]
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, seconds=0.6):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < seconds or n < 10:
        fn(); n += 1
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in sizes:
    T = (B + 63) // 64
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    w = torch.rand(B, dtype=torch.float64, device=dev)
    acc = torch.zeros(R, dtype=torch.float64, device=dev)
    root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); time.sleep(leaf.numel() * 8 / 16e9 + 0.5)
    a = timed(lambda: f.accumulate_tiled(leaf, w, acc, B))
    e = timed(lambda: f.eval_tiled(root, leaf, B))
    print(f"{name} B = {B:>9d}: accumulate {B / a / 1e3:8.1f} Msamples/s (frac {8 * L * B / a / 1e6 / 8000:.3f})   evaluate {B / e / 1e3:8.1f} Mevals/s (frac {8 * (L + R) * B / e / 1e6 / 8000:.3f})", flush=True)
    nb = leaf.numel() * 8
    del leaf, w, root
    torch.cuda.empty_cache(); torch.cuda.synchronize(); time.sleep(nb / 16e9 + 0.5)
