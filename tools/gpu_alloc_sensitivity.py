"""GPU dev tool (round 5): do the latency-bound kernels (one or two waves per SIMD, 0.5-0.6 of a roof) depend on how their batch is backed?
Per workload, several rounds of: tile-major leaves from torch.empty / one VMM allocation / 2 MB VMM chunks / 1 GB chunks, roots from torch;
evaluation timed after a pause that lets the driver's wipe of the previous round's memory finish.
usage: gpu_alloc_sensitivity.py [workloads] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["parquet_sigma5", "parquet_sigma4_insdyn", "gv_sigma5"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
st = torch.cuda.current_stream().cuda_stream
B = 2_000_000


def timed(fn, n=20, warm=40):
    for _ in range(warm): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return sum(ev[k].elapsed_time(ev[k + 1]) for k in range(n)) / n


for name in names:
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa"); h = f.handle
    T = (B + 63) // 64
    nb = T * L * 512
    for r in range(rounds):
        for pol in ("torch", "whole", "2", "1024"):
            pad = torch.empty((37 + 211 * r) << 20, dtype=torch.uint8, device=dev)
            if pol == "torch":
                keep = torch.empty(nb, dtype=torch.uint8, device=dev); lp = keep.data_ptr()
            else:
                keep = None; lp = capi.batch_alloc(nb, 0 if pol == "whole" else int(pol) << 20)
            root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
            capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
            torch.cuda.synchronize(); time.sleep(nb / 16e9 + 0.3)
            ms = timed(lambda: h.eval_device_tiled(lp, 1, 64, 64 * L, root.data_ptr(), 1, 64, 64 * R, B, st))
            print(f"{name:24s} round {r} {pol:>6s}: {ms:7.3f} ms  {B / ms / 1e3:8.1f} Mevals/s  frac_hbm {8 * (L + R) * B / ms / 1e6 / 8000:.3f}  ({f.kernel_info()['last_kernel']})", flush=True)
            if keep is None: capi.batch_free(lp)
            del keep, root, pad
            torch.cuda.empty_cache()
            torch.cuda.synchronize(); time.sleep(nb / 16e9 + 0.3)
