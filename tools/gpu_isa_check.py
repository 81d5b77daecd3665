"""GPU check + timing of the ISA back end vs oracle (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
from feynmandiagram_jl_amd.nodetable import NodeTable
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def table(name):
    if name.startswith("gv_"): return NodeTable.load(os.path.join(GOLD, name + ".npz"))
    return workloads.get(name)
def leaves(B, L, layout, seed=11):
    if layout == "leaf_major": leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    else: leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), seed, 0, torch.cuda.current_stream().cuda_stream)
    return leaf
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["sigma2", "synthetic_small", "sigma4_standin", "gv_sigma5"]
check = "--nocheck" not in sys.argv
timeonly = "--timeonly" in sys.argv
opt = {}
for a in sys.argv:
    if a.startswith("--opt="):
        for kv in a[6:].split(","):
            k, v = kv.split("="); opt[k] = int(v)
spec_kw = dict(specialize="isa", opt=opt or None)
for name in names:
    t = table(name)
    st = t.stats()
    for layout in (() if timeonly else ("leaf_major", "sample_major")):
        for B in (1, 63, 64, 1000, 20000) + ((600_007,) if layout == 'leaf_major' else ()):   # the last: several tiles per persistent wave
            f = fd.compile_table(t, **spec_kw)
            leaf = leaves(B, t.n_leaf, layout)
            root = torch.full((B, t.n_root), -3.0, dtype=torch.float64, device=dev)
            f(root, leaf); torch.cuda.synchronize()
            if check:
                want = oracle.eval_static(t, leaf.cpu().numpy(), np.full((B, t.n_root), -3.0))
                got = root.cpu().numpy()
                ok = np.array_equal(got, want)
                print(name, layout, B, "exact" if ok else f"MISMATCH max|d|={np.abs(got-want).max()} nbad={(got!=want).sum()}", flush=True)
                if not ok: print(got[:3], want[:3])
    # timing
    for layout in ("leaf_major",):
        f = fd.compile_table(t, **spec_kw)
        B = max(1 << 14, min(1 << 22, int(2e9 / (8 * t.n_leaf))))
        leaf = leaves(B, t.n_leaf, layout)
        root = torch.empty((B, t.n_root), dtype=torch.float64, device=dev)
        for _ in range(2): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); n = 5
        for _ in range(n): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ev = B / ms * 1e3
        print(f"TIME {name} {layout} B={B} {ms:.3f} ms  {ev:.3e} evals/s  alg {ev*st['bytes_alg']/1e9:.1f} GB/s  {ev*st['flops_alg']/1e12:.2f} TFLOP/s  info={f.info()['spec_vgpr']},{f.info()['spec_lds_bytes']}", flush=True)
