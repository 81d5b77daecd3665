"""GPU dev tool: rate of the per-type kernels (fdg_spec_typed: Float32 / ComplexF64 / ComplexF32) next to the Float64 ISA kernel,
leaf-major batches, bit-checked against the typed twin on a sample.  python tools/gpu_typed_rate.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads
dev = torch.device("cuda:0")
TD = {"Float64": torch.float64, "Float32": torch.float32, "ComplexF64": torch.complex128, "ComplexF32": torch.complex64}
for name in sys.argv[1:] or ["parquet_sigma4", "gv_sigma4", "parquet_sigma4_taylor2"]:
    t = workloads.get(name)
    f = fd.compile_table(t, specialize="isa")
    B = 4_000_000
    for dtype, td in TD.items():
        x = torch.rand((t.n_leaf, B), dtype=torch.float64, device=dev)
        leaf = (torch.complex(x, torch.rand_like(x) - 0.6) if td.is_complex else x).to(td).t()
        del x
        root = torch.empty((t.n_root, B), dtype=td, device=dev).t()
        for _ in range(5): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        n = 1024
        h = np.ascontiguousarray(leaf[:n].cpu().numpy())
        want = oracle.eval_static_typed(t, h, dtype) if dtype != "Float64" else oracle.eval_static(t, h)
        ok = np.array_equal(np.ascontiguousarray(root[:n].cpu().numpy()).view(np.uint8), np.ascontiguousarray(want).view(np.uint8))
        bytes_eval = (t.n_leaf + t.n_root) * leaf.element_size()
        print(f"{name:24s} {dtype:10s} leaf-major  {'exact' if ok else 'MISMATCH'}  {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s  {bytes_eval * B / ms / 1e6:.0f} GB/s  {getattr(f, 'last_typed_kernel', None) if dtype != 'Float64' else f.kernel_info()['last_kernel']}", flush=True)
        if dtype == "ComplexF64":       # rows of (re, im) pairs: the graph spelled out on real parts through the Float64 assembly kernels
            lrm = leaf.contiguous()
            rrm = torch.empty((B, t.n_root), dtype=td, device=dev)
            for _ in range(5): f(rrm, lrm)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20): f(rrm, lrm)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            ok = np.array_equal(np.ascontiguousarray(rrm[:n].cpu().numpy()).view(np.uint8), np.ascontiguousarray(want).view(np.uint8))
            print(f"{name:24s} {dtype:10s} row-major   {'exact' if ok else 'MISMATCH'}  {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s  {bytes_eval * B / ms / 1e6:.0f} GB/s  {f.last_typed_kernel}", flush=True)
            del lrm, rrm
        del leaf, root
