"""GPU dev tool: which backing of the 70 GB batch gives which rate?  The headline launch (leaf-major and tile-major, evaluation
and fused accumulation) over batches allocated by hipMalloc (through torch) and by fdg_batch_alloc with physical chunks of
2 MB ... the whole batch, several rounds each with a pad of another size allocated first (moves the driver's allocator state).
A policy "a/b" backs the leaves by a and the roots by b (does the evaluation rate follow where the 3.2 GB of roots land?).
usage: gpu_placement_probe.py [workload] [B] [rounds] [policies, comma separated: malloc,whole,1024,32,2 (MB) or leaf/root pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
policies = (sys.argv[4] if len(sys.argv) > 4 else "malloc,whole,1024,32,2").split(",")
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
Bp = 64 * T
shift_mb = [0, 517, 3, 2051, 64, 9000, 130, 1]
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros(R, dtype=torch.float64, device=dev)


def timed(fn, n=8, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


class Buf:
    def __init__(self, nbytes, policy):
        self.policy = policy
        t0 = time.time()
        if policy == "malloc":
            self.t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.ptr = self.t.data_ptr()
        else:
            self.ptr = capi.batch_alloc(nbytes, 0 if policy == "whole" else int(policy) << 20)
        self.dt = time.time() - t0

    def free(self):
        if self.policy == "malloc":
            del self.t
            torch.cuda.empty_cache()
        else:
            capi.batch_free(self.ptr)


for pol in policies:
    for r in range(rounds):
        pad = torch.empty(max(1, shift_mb[r % len(shift_mb)]) << 20, dtype=torch.uint8, device=dev)
        lp, _, rp = pol.partition("/")
        leaf = Buf(8 * L * Bp, lp); root = Buf(8 * R * Bp, rp or lp)
        out = []
        for lay in ("leaf-major", "tile-major"):
            if lay == "leaf-major":
                capi.fill_uniform_device(leaf.ptr, B, L, 1, Bp, 1234, 0, st)
                ev = lambda: h.eval_device(leaf.ptr, 1, Bp, root.ptr, 1, Bp, B, st)
                ac = lambda: h.accumulate_device(leaf.ptr, 1, Bp, 0, acc.data_ptr(), B, st)
            else:
                capi.fill_uniform_device_tiled(leaf.ptr, B, L, 1, 64, 64 * L, 1234, 0, st)
                ev = lambda: h.eval_device_tiled(leaf.ptr, 1, 64, 64 * L, root.ptr, 1, 64, 64 * R, B, st)
                ac = lambda: h.accumulate_device_tiled(leaf.ptr, 1, 64, 64 * L, 0, acc.data_ptr(), B, st)
            e, a = timed(ev), timed(ac)
            out.append(f"{lay}: eval {e:.3f} ms frac {8 * (L + R) * B / e / 1e6 / 8000:.3f}  acc {a:.3f} ms frac {8 * L * B / a / 1e6 / 8000:.3f}")
        print(f"{pol:>13} round {r} pad {shift_mb[r % len(shift_mb)]:5d} MB  alloc {leaf.dt:.2f}+{root.dt:.2f} s  leaf @ {leaf.ptr:#x}  " + "  |  ".join(out), flush=True)
        leaf.free(); root.free()
        del pad
        torch.cuda.empty_cache()
