"""GPU dev tool (round 5): what would a one-wave kernel run at if its leaf loads hit the cache?  The same kernel over the same number of tiles, with the
tiles' leaf blocks OVERLAPPING (tile stride of a few doubles instead of 64 L): the whole "batch" is a few MB, L2 / Infinity-Cache resident.  Results are
garbage; the roots are written to a real batch.   usage: gpu_resident_probe.py workload B [tile strides in doubles ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
strides = [int(x) for x in sys.argv[3:]] or [0]
t = workloads.get(name); L, R = t.n_leaf, t.n_root
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
f = fd.compile_table(t, specialize="isa")
h = f.handle
leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
for lts in strides:
    real = lts == 0
    s = 64 * L if real else lts
    run = lambda: h.eval_device_tiled(leaf.data_ptr(), 1, 64, s, root.data_ptr(), 1, 64, 64 * R, B, st)
    for _ in range(30): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    foot = (T * s + 64 * L) * 8 / 1e6
    print(f"{name} leaf tile stride {s:6d} doubles ({'the real batch' if real else 'overlapping tiles'}; leaf footprint {foot:9.1f} MB) {f.kernel_info()['last_kernel']:18s} {ms:7.3f} ms  {B / ms / 1e3:8.1f} Mevals/s", flush=True)
