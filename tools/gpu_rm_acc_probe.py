"""GPU dev tool: fused accumulation over a row-major [B, L] batch (compile_Python's layout) against evaluation of the same batch and
against both on a tile-major batch.  usage: gpu_rm_acc_probe.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=10, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name in sys.argv[1:] or ["parquet_sigma4", "sigma2", "parquet_sigma3", "gv_sigma4", "gv_sigma4_taylor2"]:
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    B = max(1 << 14, min(16_000_000, int(2.4e9 / (8 * L)))) // 64 * 64
    f = fd.compile_table(t, specialize="isa")
    rm = torch.empty((B, L), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(rm.data_ptr(), B, L, L, 1, 11, 0, st)
    root = torch.empty((B, R), dtype=torch.float64, device=dev)
    acc = torch.zeros(R, dtype=torch.float64, device=dev)
    e = timed(lambda: f(root, rm)); ke = f.handle.kernel_info()["last_kernel"]
    a = timed(lambda: f.accumulate(rm, None, acc)); ka = f.handle.kernel_info()["last_kernel"]
    tm = torch.empty((B // 64, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(tm.data_ptr(), B, L, 1, 64, 64 * L, 11, 0, st)
    rt = torch.empty((B // 64, R, 64), dtype=torch.float64, device=dev)
    e2 = timed(lambda: f.eval_tiled(rt, tm, B))
    a2 = timed(lambda: f.accumulate_tiled(tm, None, acc, B))
    fe = lambda ms, by: by * B / ms / 1e6 / 8000
    print(f"{name:24s} L={L:4d} B={B}  row-major eval {fe(e, 8 * (L + R)):.3f} [{ke}]  acc {fe(a, 8 * L):.3f} [{ka}]   tile-major eval {fe(e2, 8 * (L + R)):.3f}  acc {fe(a2, 8 * L):.3f}", flush=True)
