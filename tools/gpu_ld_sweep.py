"""GPU dev tool: the evaluator on a column-major B x L leaf matrix whose leading dimension is B + pad (a Julia view of a taller
matrix; LAPACK's LDA): how much does the column stride matter to the HBM channels?  python tools/gpu_ld_sweep.py WORKLOAD B pad ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
for rep in range(2):
    for pad in [int(x) for x in sys.argv[3:]]:
        ld = B + pad
        leaf = torch.empty((L, ld), dtype=torch.float64, device=dev)[:, :B].t()
        root = torch.empty((R, ld), dtype=torch.float64, device=dev)[:, :B].t()
        capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 11, 0, torch.cuda.current_stream().cuda_stream)
        f(root, leaf); torch.cuda.synchronize()
        ok = np.array_equal(root[:2048].cpu().numpy(), oracle.eval_static(t, leaf[:2048].cpu().numpy()))
        for _ in range(10): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(15): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 15
        print(f"{name} B={B} ld=B+{pad}: {'exact' if ok else 'MISMATCH'} {ms:.3f} ms {B / ms * 1e3:.3e} evals/s {B / ms * 1e3 * 8 * (L + R) / 1e9:.0f} GB/s", flush=True)
        del leaf, root
        torch.cuda.empty_cache()
