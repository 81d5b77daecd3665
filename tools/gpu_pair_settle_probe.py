"""GPU dev tool (round 5): after fdg_batch_alloc_pair returns, do the rates of the batch drift?  Evaluation and fused accumulation of the
whole batch, timed every half second for `secs` seconds.   usage: gpu_pair_settle_probe.py [secs] [extra_flags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
xf = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = 100_000_000
t = workloads.get("parquet_sigma4"); L, R = t.n_leaf, t.n_root
h = fd.compile_table(t, specialize="isa").handle
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros(R, dtype=torch.float64, device=dev)


def timed(fn, n=4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    fn(); ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


t0 = time.time()
lp, rp, info = capi.batch_alloc_pair(h, B, 0, True, False, xf)
t1 = time.time()
capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 1234, 0, st)
print(f"allocated in {t1 - t0:.2f} s: mapped pairs mean {info['gbs_after_mean'] / 8000:.3f} min {info['gbs_after_min'] / 8000:.3f}, fillers {info['n_filler']}, candidates {info['n_candidate']}", flush=True)
while time.time() - t1 < secs:
    e = timed(lambda: h.eval_device_tiled(lp, 1, 64, 64 * L, rp, 1, 64, 64 * R, B, st))
    a = timed(lambda: h.accumulate_device_tiled(lp, 1, 64, 64 * L, 0, acc.data_ptr(), B, st))
    print(f"  t = {time.time() - t1:5.2f} s: eval {8 * (L + R) * B / e / 1e6 / 8000:.3f}  acc {8 * L * B / a / 1e6 / 8000:.3f}", flush=True)
    time.sleep(0.4)
