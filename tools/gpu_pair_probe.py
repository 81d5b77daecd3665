"""GPU dev tool (round 5): does the rate of a piece of the headline batch belong to the LEAF pages, the ROOT pages, or the PAIR?
One allocation of the 70 GB tile-major leaves and of the roots; every `step`-th 2 GB leaf segment is evaluated into every `step`-th root piece
(the piece another segment's roots would go to): a matrix of fractions of 8 TB/s.  Rows that are uniformly slow: the leaf pages; columns: the
root pages; a pattern that depends on both: leaf and root pages interfere pairwise (e.g. the same DRAM banks).  Then the fused accumulation
(no root written) per leaf segment.
usage: gpu_pair_probe.py [workload] [B] [policy] [step] [n_alloc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
pol = sys.argv[3] if len(sys.argv) > 3 else "malloc"
step = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n_alloc = int(sys.argv[5]) if len(sys.argv) > 5 else 2
t = workloads.get(name); L, R = t.n_leaf, t.n_root
h = fd.compile_table(t, specialize="isa").handle
T = (B + 63) // 64
Bp = 64 * T
st = torch.cuda.current_stream().cuda_stream
seg_tiles = (2 << 30) // (512 * L)
n_seg = T // seg_tiles
acc = torch.zeros(R, dtype=torch.float64, device=dev)


def timed(fn, n=5, warm=1):
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


for a in range(n_alloc):
    lp, kl = alloc(8 * L * Bp, pol)
    rp, kr = alloc(8 * R * Bp, "malloc")
    rp2, kr2 = alloc(8 * R * Bp, "malloc")           # a second root buffer, elsewhere
    capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
    n = seg_tiles * 64
    segs = list(range(0, n_seg, step))
    print(f"allocation {a} ({pol}): leaf @ {lp:#x} root @ {rp:#x} root2 @ {rp2:#x}; rows = leaf segment, columns = root piece of segment {segs}, then root2 pieces", flush=True)
    for s in segs:
        row = []
        for rbase in (rp, rp2):
            for q in segs:
                ms = timed(lambda: h.eval_device_tiled(lp + s * seg_tiles * 512 * L, 1, 64, 64 * L, rbase + q * seg_tiles * 512 * R, 1, 64, 64 * R, n, st))
                row.append(8 * (L + R) * n / ms / 1e6 / 8000)
        ms = timed(lambda: h.accumulate_device_tiled(lp + s * seg_tiles * 512 * L, 1, 64, 64 * L, 0, acc.data_ptr(), n, st))
        print(f"  leaf seg {s:2d}: " + " ".join(f"{x:.3f}" for x in row[:len(segs)]) + "  |  " + " ".join(f"{x:.3f}" for x in row[len(segs):]) + f"  | acc {8 * L * n / ms / 1e6 / 8000:.3f}", flush=True)
    if kl is None: capi.batch_free(lp)
    del kl, kr, kr2
    torch.cuda.empty_cache()
