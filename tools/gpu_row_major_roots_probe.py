"""GPU dev tool: leaf-major leaves and ROW-MAJOR roots [B, R] (a torch caller's natural root tensor) on graphs with many roots: the root
scratch detour (FDG_ROOT_SCRATCH_MIN, default 16 roots) against direct stores.  usage: gpu_row_major_roots_probe.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for name in sys.argv[1:] or ["parquet_ver4_4", "parquet_ver4_3", "gv_ver4_4"]:
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    B = max(1 << 14, min(4_000_000, int(1.6e9 / (8 * L)))) // 64 * 64 + 37
    f = fd.compile_table(t, specialize="isa")
    lm = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(lm.data_ptr(), B, L, 1, B, 11, 0, st)
    rr = torch.full((B, R), 7.0, dtype=torch.float64, device=dev)
    rc = torch.empty((R, B), dtype=torch.float64, device=dev).t()
    f(rr, lm); f(rc, lm); torch.cuda.synchronize()
    same = bool(torch.equal(rr, rc))
    def timed(fn, n=8):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    a = timed(lambda: f(rr, lm)); ka = f.kernel_info()["last_kernel"]
    b = timed(lambda: f(rc, lm))
    print(f"{name:18s} L={L} R={R} B={B}  row-major roots {B / a * 1e3:.3e}/s [{ka}]  column-major roots {B / b * 1e3:.3e}/s  same bits {same}", flush=True)
