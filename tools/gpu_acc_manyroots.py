"""GPU dev tool: evaluation against weighted accumulation on the graphs with many roots (the vertex functions of
example/benchmark.jl: 180 roots, and example/benchmark_GV.jl: 26 roots), both layouts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads
dev = torch.device("cuda:0")
for name, B in (("parquet_ver4_4", 500_000), ("gv_ver4_4", 500_000), ("parquet_sigma4_taylor2", 4_000_000)):
    t = workloads.get(name)
    f = fd.compile_table(t, specialize="isa")
    for layout in ("leaf_major", "sample_major"):
        leaf = torch.rand((t.n_leaf, B), dtype=torch.float64, device=dev).t() if layout == "leaf_major" else torch.rand((B, t.n_leaf), dtype=torch.float64, device=dev)
        root = torch.empty((t.n_root, B), dtype=torch.float64, device=dev).t() if layout == "leaf_major" else torch.empty((B, t.n_root), dtype=torch.float64, device=dev)
        w = torch.rand(B, dtype=torch.float64, device=dev)
        for what in ("eval", "acc"):
            fn = (lambda: f(root, leaf)) if what == "eval" else (lambda: f.accumulate(leaf, w))
            for _ in range(3): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(f"{name:24s} {layout:12s} {what:4s} {dt*1e3:8.3f} ms  {B/dt:.3e} samples/s  kernel {f.kernel_info()['last_kernel']}", flush=True)
