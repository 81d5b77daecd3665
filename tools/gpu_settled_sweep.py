"""Settled-state timing of optimizer parameter sets (80 warm-up + 100 timed launches each; dev tool).
usage: python tools/gpu_settled_sweep.py workload "k=v,k=v" ["k=v,..." ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
name = sys.argv[1]
t = workloads.get(name)
B = {"sigma2": 64_000_000, "gv_sigma4": 8_000_000, "gv_sigma5": 2_000_000, "gv_sigma6": 500_000}.get(name, 4_000_000)
dev = torch.device("cuda:0")
leaf = torch.empty((t.n_leaf, B), dtype=torch.float64, device=dev).t()
capi.fill_uniform_device(leaf.data_ptr(), B, t.n_leaf, leaf.stride(0), leaf.stride(1), 1234, 0, torch.cuda.current_stream().cuda_stream)
root = torch.empty((t.n_root, B), dtype=torch.float64, device=dev).t()
for spec in sys.argv[2:]:
    opt = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))} if spec != "tuned" else None
    f = fd.compile_table(t, specialize="isa", opt=opt, cache_dir="/tmp/kc_sweep" if opt else None)
    for _ in range(80): f(root, leaf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f(root, leaf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print(f"{name} {spec}: {ms:.4f} ms  {B/ms*1e3:.3e} evals/s", flush=True)
