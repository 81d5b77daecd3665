"""GPU dev tool (round 5): leaf segments of one plain allocation x small, separately drawn root chunks (VMM, `root_mb` MB each): the matrix
of evaluation rates as characters (F >= 0.83, m 0.79-0.83, s < 0.79 of 8 TB/s) -- how many kinds of root chunks are there, at what granularity?
usage: gpu_pair_matrix2.py [workload] [B] [seg_mb] [root_mb] [n_cand] [leaf policy]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
seg_mb = int(sys.argv[3]) if len(sys.argv) > 3 else 1008
root_mb = int(sys.argv[4]) if len(sys.argv) > 4 else 48
n_cand = int(sys.argv[5]) if len(sys.argv) > 5 else 64
pol = sys.argv[6] if len(sys.argv) > 6 else "malloc"
t = workloads.get(name); L, R = t.n_leaf, t.n_root
h = fd.compile_table(t, specialize="isa").handle
T = (B + 63) // 64
Bp = 64 * T
st = torch.cuda.current_stream().cuda_stream
seg_tiles = (seg_mb << 20) // (512 * L)
assert seg_tiles * 512 * R <= root_mb << 20
n_seg = T // seg_tiles


def timed(fn, n=3, warm=1):
    for _ in range(warm): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


def ch(x): return "F" if x >= 0.83 else ("m" if x >= 0.79 else "s")


if pol == "malloc":
    keep = torch.empty(8 * L * Bp, dtype=torch.uint8, device=dev); lp = keep.data_ptr()
else:
    lp = capi.batch_alloc(8 * L * Bp, 0 if pol == "whole" else int(pol) << 20)
cands = []
for j in range(n_cand):
    cands.append(capi.batch_alloc(root_mb << 20, 0))
    if j % 8 == 7: cands.append(None)                      # a 1 GB filler every eight candidates moves the allocator along
fillers = [capi.batch_alloc(1 << 30, 0) for c in cands if c is None]
cands = [c for c in cands if c is not None]
capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
n = seg_tiles * 64
print(f"{n_seg} leaf segments of {seg_mb} MB ({pol}) x {len(cands)} root chunks of {root_mb} MB (a 1 GB filler drawn after every eighth)", flush=True)
for s in range(n_seg):
    row = []
    for c in cands:
        ms = timed(lambda: h.eval_device_tiled(lp + s * seg_tiles * 512 * L, 1, 64, 64 * L, c, 1, 64, 64 * R, n, st))
        row.append(8 * (L + R) * n / ms / 1e6 / 8000)
    print(f"  {s:2d} " + "".join(ch(x) for x in row) + f"  max {max(row):.3f} min {min(row):.3f}", flush=True)
