import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
from feynmandiagram_jl_amd.nodetable import NodeTable
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
name = sys.argv[1]
t = NodeTable.load(os.path.join(GOLD, name + ".npz")) if name.startswith("gv_") else workloads.get(name)
st = t.stats()
f = fd.compile_table(t, specialize="isa")
for layout in ("leaf_major", "sample_major"):
    for B in [int(x) for x in sys.argv[2].split(",")]:
        L = t.n_leaf
        leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t() if layout == "leaf_major" else torch.empty((B, L), dtype=torch.float64, device=dev)
        capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 5, 0, torch.cuda.current_stream().cuda_stream)
        root = torch.empty((B, t.n_root), dtype=torch.float64, device=dev)
        for _ in range(2): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); n = 5
        for _ in range(n): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ev = B / ms * 1e3
        print(f"{name} {layout} B={B} {ms:.3f} ms {ev:.3e} evals/s {ev*st['flops_alg']/1e12:.2f} TF  {ev*st['bytes_alg']/1e9:.0f} GB/s", flush=True)
        del leaf, root
