"""GPU dev tool: within ONE allocation of the headline batch, does the evaluation rate depend on where inside its buffer the roots start?
(The rate's two modes follow the allocation; if they followed the roots' offset modulo some interleaving unit, a fixed offset could select
the fast one.)  Tile-major batch; the roots' base is moved by the listed offsets inside a buffer with 2 GB of slack.
usage: gpu_root_offset_probe.py [workload] [B] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
h = f.handle
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
OFFS = [0, 128, 512, 2048, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, 256 << 20, 1 << 30, (1 << 30) + (1 << 20), 0]


def timed(fn, n=6, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(n))


for r in range(rounds):
    pad = torch.empty(max(1, [0, 517, 3, 2051][r % 4]) << 20, dtype=torch.uint8, device=dev)
    leaf = torch.empty(8 * L * 64 * T, dtype=torch.uint8, device=dev)
    rootbuf = torch.empty(8 * R * 64 * T + (2 << 30), dtype=torch.uint8, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    out = []
    for off in OFFS:
        rp = rootbuf.data_ptr() + off
        ms = timed(lambda: h.eval_device_tiled(leaf.data_ptr(), 1, 64, 64 * L, rp, 1, 64, 64 * R, B, st))
        out.append(f"{off:>11d}: {8 * (L + R) * B / ms / 1e6 / 8000:.3f}")
    print(f"round {r}  leaf @ {leaf.data_ptr():#x} roots @ {rootbuf.data_ptr():#x}\n   " + "\n   ".join(out), flush=True)
    del leaf, rootbuf, pad
    torch.cuda.empty_cache()
