"""GPU dev tool (round 5): the tile-major immediate-offset variant (fdg_isa_eval_tm) against the streaming one (fdg_isa_eval_nt) on the same
batch, same process (handle option FDG_ISA_NO_TM), and a bitwise comparison of their roots.   usage: gpu_tm_ab.py [workloads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["parquet_sigma4_insdyn", "parquet_sigma5", "parquet_ver4_4"]
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=10, warm=20):
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
    B = {"parquet_ver4_4": 512_000, "gv_ver4_4": 512_000}.get(name, 2_000_000) + 37
    T = (B + 63) // 64
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    out = {}
    for tag, opt in (("tm", None), ("nt", "1"), ("tm", None), ("nt", "1")):
        h.set_option("FDG_ISA_NO_TM", opt)
        root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
        ms = timed(lambda: f.eval_tiled(root, leaf, B))
        k = f.kernel_info()["last_kernel"]
        print(f"{name:24s} {k:22s} {ms:8.3f} ms  {B / ms / 1e3:9.1f} Mevals/s", flush=True)
        out[tag] = root
    same = torch.equal(out["tm"], out["nt"])
    if not same:
        d = (out["tm"] != out["nt"]).nonzero()
        print(f"   DIFFER at {d.shape[0]} places; first {d[:5].tolist()}; tiles {sorted(set(d[:, 0].tolist()))[:10]} ...", flush=True)
    else:
        print("   same bits", flush=True)
