"""GPU dev tool (round 6): the headline batch (tile-major, fdg_batch_alloc_pair) under root-store cache policies, A B A B on ONE paired batch.
usage: gpu_root_policy_paired.py workload B policy1 policy2 ...   ("-" = default)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f0 = fd.compile_table(t, specialize="isa")
pb = f0.tile_major_pair(B, dev, calibrate=True)
print("paired batch:", {k: pb.info[k] for k in ("n_chunk", "n_matched", "level_reached", "seconds")}, flush=True)
st = torch.cuda.current_stream().cuda_stream
capi.fill_uniform_device_tiled(pb.leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
fs = []
for pol in sys.argv[3:]:
    opts = {} if pol == "-" else {"FDG_ISA_ROOT_POLICY": pol}
    fs.append((pol, fd.compile_table(t, specialize="isa", options=opts, cache_dir="/tmp/sweep_cache")))
ref = None
for rep in range(3):
    for pol, f in fs:
        for _ in range(30): f.eval_tiled(pb.root, pb.leaf, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for _ in range(n): f.eval_tiled(pb.root, pb.leaf, B)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if ref is None: ref = pb.root[:64].clone()
        print(f"{name} paired B={B} root policy [{pol}] {ms:7.3f} ms frac_hbm {8 * (L + R) * B / ms / 1e6 / 8000:.4f} same bits {bool(torch.equal(pb.root[:64], ref))}", flush=True)
pb.free()
