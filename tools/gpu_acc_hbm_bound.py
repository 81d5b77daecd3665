import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
for name, B in (("parquet_sigma4", 100_000_000), ("gv_sigma4", 16_000_000), ("sigma2", 64_000_000), ("parquet_sigma4_dyn", 8_000_000)):
    t = workloads.get(name)
    f = fd.compile_table(t, specialize="isa")
    leaf = torch.empty((t.n_leaf, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(leaf.data_ptr(), B, t.n_leaf, leaf.stride(0), leaf.stride(1), 1234, 0, torch.cuda.current_stream().cuda_stream)
    w = torch.rand(B, dtype=torch.float64, device=dev)
    root = torch.empty((t.n_root, B), dtype=torch.float64, device=dev).t()
    for what in ("eval", "acc", "acc_unit"):
        fn = {"eval": lambda: f(root, leaf), "acc": lambda: f.accumulate(leaf, w), "acc_unit": lambda: f.accumulate(leaf, None)}[what]
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        by = 8 * (t.n_leaf + (t.n_root if what == "eval" else (1 if what == "acc" else 0)))
        print(f"{name:20s} {what:8s} {ms:8.3f} ms {B/ms*1e3:.3e} /s  {by*B/ms/1e6:.0f} GB/s ({by*B/ms/1e6/8000:.3f} of 8 TB/s)  kernel {f.kernel_info()['last_kernel']}", flush=True)
    del leaf, w, root
    torch.cuda.empty_cache()
