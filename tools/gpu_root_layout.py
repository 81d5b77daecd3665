"""Row-major [B, R] vs column-major (Julia B x R) roots with leaf-major leaves, settled clocks (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
for name, B in (("gv_sigma4_taylor2", 4_000_000), ("gv_sigma4", 8_000_000), ("sigma2", 64_000_000), ("gv_sigma5", 2_000_000)):
    t = workloads.get(name)
    leaf = torch.empty((t.n_leaf, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(leaf.data_ptr(), B, t.n_leaf, leaf.stride(0), leaf.stride(1), 1234, 0, torch.cuda.current_stream().cuda_stream)
    f = fd.compile_table(t, specialize="isa")
    for lay in ("row-major", "column-major", "row-major", "column-major"):
        root = torch.empty((B, t.n_root), dtype=torch.float64, device=dev) if lay == "row-major" else torch.empty((t.n_root, B), dtype=torch.float64, device=dev).t()
        for _ in range(80): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"{name} roots {lay}: {ms:.4f} ms {B/ms*1e3:.3e} evals/s", flush=True)
    del leaf, root
