"""GPU dev tool: the row-major variant with the matrix's own row pitch (L doubles) and with rows padded to whole cache lines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name = sys.argv[1]; B = int(sys.argv[2])
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
for pitch in (L, (L + 15) // 16 * 16):
    for rp in (R, 16):
        leaf = torch.empty((B, pitch), dtype=torch.float64, device=dev)[:, :L]
        root = torch.empty((B, rp), dtype=torch.float64, device=dev)[:, :R]
        capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 11, 0, torch.cuda.current_stream().cuda_stream)
        f(root, leaf); torch.cuda.synchronize()
        ok = np.array_equal(root[:2048].cpu().numpy(), oracle.eval_static(t, leaf[:2048].cpu().numpy()))
        for _ in range(10): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name} row pitch {pitch} root pitch {rp}: {'exact' if ok else 'MISMATCH'} {ms:.3f} ms {B / ms * 1e3:.3e} evals/s", flush=True)
