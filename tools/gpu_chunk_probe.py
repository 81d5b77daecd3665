"""GPU dev tool (round 5): is the rate of the headline evaluation a LOCAL property of the pages under the batch?
The 70 GB tile-major batch is allocated under a backing policy, the whole batch is timed, and then every segment of `seg_gb` GB of
the leaves (with the matching piece of the roots) is timed on its own, twice.  If the segments of one allocation differ from each
other and keep their rates from pass to pass, a slow allocation is a batch with slow pieces and an allocator can probe and re-draw them;
if they all run at the allocation's rate, the rate is a property of the whole mapping.
usage: gpu_chunk_probe.py [workload] [B] [policies: malloc,whole,1024,32,2] [seg_gb] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
policies = (sys.argv[3] if len(sys.argv) > 3 else "malloc,whole,1024").split(",")
seg_gb = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
Bp = 64 * T
st = torch.cuda.current_stream().cuda_stream
seg_tiles = int(seg_gb * (1 << 30)) // (512 * L)
n_seg = T // seg_tiles


def timed(fn, n=6, warm=2):
    for _ in range(warm): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


def alloc(nbytes, policy):
    if policy == "malloc":
        x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return x.data_ptr(), x
    return capi.batch_alloc(nbytes, 0 if policy == "whole" else int(policy) << 20), None


def frac(ms, n): return 8 * (L + R) * n / ms / 1e6 / 8000


for pol in policies:
    for r in range(rounds):
        lp, keep_l = alloc(8 * L * Bp, pol)
        rp, keep_r = alloc(8 * R * Bp, "malloc")
        capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
        whole = timed(lambda: h.eval_device_tiled(lp, 1, 64, 64 * L, rp, 1, 64, 64 * R, B, st), n=8, warm=5)
        passes = []
        for p in range(2):
            fr = []
            for s in range(n_seg):
                lo = s * seg_tiles
                ms = timed(lambda: h.eval_device_tiled(lp + lo * 512 * L, 1, 64, 64 * L, rp + lo * 512 * R, 1, 64, 64 * R, seg_tiles * 64, st), n=5, warm=1)
                fr.append(frac(ms, seg_tiles * 64))
            passes.append(fr)
        a, b = torch.tensor(passes[0]), torch.tensor(passes[1])
        corr = float(torch.corrcoef(torch.stack([a, b]))[0, 1]) if n_seg > 2 else float("nan")
        print(f"{pol:>7} round {r} leaf @ {lp:#x}: whole batch {frac(whole, B):.3f} | {n_seg} segments of {seg_gb} GB: min {a.min():.3f} median {a.median():.3f} "
              f"max {a.max():.3f} mean {a.mean():.3f}; pass 2 mean {b.mean():.3f}; corr(pass 1, pass 2) {corr:.2f}", flush=True)
        print("   pass 1: " + " ".join(f"{x:.3f}" for x in passes[0]))
        print("   pass 2: " + " ".join(f"{x:.3f}" for x in passes[1]), flush=True)
        if keep_l is None: capi.batch_free(lp)
        del keep_l, keep_r
        torch.cuda.empty_cache()
