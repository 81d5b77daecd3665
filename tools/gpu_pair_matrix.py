"""GPU dev tool (round 5): the full leaf-segment x root-piece matrix of one allocation as characters (F >= 0.83, m 0.79-0.83, s < 0.79 of
8 TB/s), plus a SELF column: the segment's roots written into the last eighth of the segment's own leaf pages (only the first 7/8 of its tiles
are evaluated) -- is a matched pair recognisable without any root candidate?
usage: gpu_pair_matrix.py [workload] [B] [policy] [seg_gb] [n_alloc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
pol = sys.argv[3] if len(sys.argv) > 3 else "malloc"
seg_gb = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
n_alloc = int(sys.argv[5]) if len(sys.argv) > 5 else 1
t = workloads.get(name); L, R = t.n_leaf, t.n_root
h = fd.compile_table(t, specialize="isa").handle
T = (B + 63) // 64
Bp = 64 * T
st = torch.cuda.current_stream().cuda_stream
seg_tiles = int(seg_gb * (1 << 30)) // (512 * L)
n_seg = T // seg_tiles


def timed(fn, n=4, warm=1):
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


def ch(x): return "F" if x >= 0.83 else ("m" if x >= 0.79 else "s")


for a in range(n_alloc):
    lp, kl = alloc(8 * L * Bp, pol)
    rp, kr = alloc(8 * R * Bp, pol if pol != "malloc" else "malloc")
    capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
    n = seg_tiles * 64
    n7 = (seg_tiles * 7 // 8) * 64
    print(f"allocation {a} ({pol}), {n_seg} segments of {seg_gb} GB: leaf @ {lp:#x} root @ {rp:#x}; rows = leaf segment, columns = root piece; then self (7/8 of the tiles, roots into the "
          f"segment's last eighth), then the same 7/8 into its own root piece", flush=True)
    for s in range(n_seg):
        lbase = lp + s * seg_tiles * 512 * L
        row = []
        for q in range(n_seg):
            ms = timed(lambda: h.eval_device_tiled(lbase, 1, 64, 64 * L, rp + q * seg_tiles * 512 * R, 1, 64, 64 * R, n, st))
            row.append(8 * (L + R) * n / ms / 1e6 / 8000)
        self_root = lbase + (seg_tiles * 7 // 8) * 512 * L
        ms = timed(lambda: h.eval_device_tiled(lbase, 1, 64, 64 * L, self_root, 1, 64, 64 * R, n7, st))
        f_self = 8 * (L + R) * n7 / ms / 1e6 / 8000
        ms = timed(lambda: h.eval_device_tiled(lbase, 1, 64, 64 * L, rp + s * seg_tiles * 512 * R, 1, 64, 64 * R, n7, st))
        f_own = 8 * (L + R) * n7 / ms / 1e6 / 8000
        print(f"  {s:2d} " + "".join(ch(x) for x in row) + f"  max {max(row):.3f} min {min(row):.3f} | self {f_self:.3f} own(7/8) {f_own:.3f} own {row[s]:.3f}", flush=True)
    if kl is None: capi.batch_free(lp); capi.batch_free(rp)
    del kl, kr
    torch.cuda.empty_cache()
