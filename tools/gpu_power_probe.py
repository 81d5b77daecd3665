"""GPU dev tool: is a workload's rate set by the power budget?  The same ISA kernel on (a) its normal leaf-major batch of random
leaves, (b) the same batch zero-filled (same HBM traffic, no toggling in the vector ALUs), (c) one random row broadcast to every
sample (sample stride 0: same arithmetic on realistic values, next to no HBM traffic), (d) a zero row broadcast.
python tools/gpu_power_probe.py WORKLOAD [B [random,zero,row]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name = sys.argv[1]
t = workloads.get(name)
L, R = t.n_leaf, t.n_root
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
f = fd.compile_table(t, specialize="isa")
h = f.handle
st = torch.cuda.current_stream().cuda_stream
leaf = torch.empty((L, B), dtype=torch.float64, device=dev)
root = torch.empty((R, B), dtype=torch.float64, device=dev)
row = torch.empty((L,), dtype=torch.float64, device=dev)

def timed(ss, ls, ptr, n=40, warm=25):
    for _ in range(warm): h.eval_device(ptr, ss, ls, root.data_ptr(), 1, B, B, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): h.eval_device(ptr, ss, ls, root.data_ptr(), 1, B, B, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["random", "zero"]
for label, fill in (("random leaves", True), ("zero leaves", False)):
    if label.split()[0] not in modes: continue
    if fill:
        capi.fill_uniform_device(leaf.data_ptr(), B, L, 1, B, 11, 0, st)
        capi.fill_uniform_device(row.data_ptr(), 1, L, L, 1, 12, 0, st)
    else:
        leaf.zero_(); row.zero_()
    ms = timed(1, B, leaf.data_ptr())
    print(f"{name} {label:14s} leaf-major batch      {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s  kernel {h.kernel_info()['last_kernel'] if hasattr(h, 'kernel_info') else ''}", flush=True)
    if "row" in modes:
        ms = timed(0, 1, row.data_ptr())
        print(f"{name} {label:14s} one row for every sample {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s", flush=True)
