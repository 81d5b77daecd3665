"""GPU dev tool (round 5): the headline batch allocated by fdg_batch_alloc_pair -- calibrated and not -- against plain hipMalloc: the
allocator's report, then the whole-batch evaluation and fused accumulation rates (fractions of 8 TB/s), several rounds each.
usage: gpu_pair_alloc_probe.py [workload] [B] [rounds] [chunk_mb] [modes: cal,nocal,malloc]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
chunk_mb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
modes = (sys.argv[5] if len(sys.argv) > 5 else "cal,nocal,malloc").split(",")
t = workloads.get(name); L, R = t.n_leaf, t.n_root
h = fd.compile_table(t, specialize="isa").handle
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros(R, dtype=torch.float64, device=dev)


def timed(fn, n=10, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
    return min(ms), sum(ms) / n, max(ms)


for r in range(rounds):
    for mode in modes:
        pad = torch.empty((517 * r + 3) << 20, dtype=torch.uint8, device=dev)       # moves the allocators along between rounds
        t0 = time.time()
        if mode == "malloc":
            leaf = torch.empty(T * L * 64, dtype=torch.float64, device=dev); root = torch.empty(T * R * 64, dtype=torch.float64, device=dev)
            lp, rp, info = leaf.data_ptr(), root.data_ptr(), {}
        else:
            lp, rp, info = capi.batch_alloc_pair(h, B, chunk_mb << 20, mode == "cal", verbose=True)
        dt = time.time() - t0
        capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
        e = timed(lambda: h.eval_device_tiled(lp, 1, 64, 64 * L, rp, 1, 64, 64 * R, B, st))
        a = timed(lambda: h.accumulate_device_tiled(lp, 1, 64, 64 * L, 0, acc.data_ptr(), B, st))
        fe = [8 * (L + R) * B / x / 1e6 / 8000 for x in e]; fa = [8 * L * B / x / 1e6 / 8000 for x in a]
        rep = ""
        if info:
            rep = (f" | chunks {info['n_chunk']} x {info['chunk_tiles']} tiles, candidates {info['n_candidate']}, fillers {info['n_filler']}, probes {info['n_probe']}, "
                   f"calibrated {info['calibrated']}, matched {info['n_matched']}; levels {info['gbs_fast'] / 8000:.3f}/{info['gbs_slow'] / 8000:.3f}; "
                   f"pairs before mean {info['gbs_before_mean'] / 8000:.3f} min {info['gbs_before_min'] / 8000:.3f} -> after mean {info['gbs_after_mean'] / 8000:.3f} "
                   f"min {info['gbs_after_min'] / 8000:.3f}; {info['seconds']:.2f} s")
        print(f"round {r} {mode:>6}: eval frac best {fe[0]:.3f} mean {fe[1]:.3f} worst {fe[2]:.3f} | acc best {fa[0]:.3f} mean {fa[1]:.3f} | alloc {dt:.2f} s{rep}", flush=True)
        if mode == "malloc":
            del leaf, root
        else:
            capi.batch_free(lp); capi.batch_free(rp)
        del pad
        torch.cuda.empty_cache()
