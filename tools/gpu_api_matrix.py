"""GPU dev tool: every (layout, mode) of the boundary on graphs of every size class -- fraction of the binding roof and the kernel that ran.
Layouts: leaf-major (Julia column-major), tile-major, row-major with contiguous rows (compile_Python), row-major with padded rows.
Modes: evaluate (roots written), accumulate (fused where the back end has it).  Finds the combinations that fall off a cliff.
usage: gpu_api_matrix.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


names = sys.argv[1:] or ["sigma2", "parquet_sigma3", "parquet_sigma4", "gv_sigma4", "parquet_sigma4_taylor2", "gv_sigma5", "parquet_ver4_4", "gv_ver4_4"]
for name in names:
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    s = t.stats()
    B = max(1 << 14, min(8_000_000, int(1.6e9 / (8 * L)))) // 64 * 64
    f = fd.compile_table(t, specialize="isa")
    ops = f.kernel_info().get("n_valu", [0])[0] if isinstance(f.kernel_info().get("n_valu"), (list, tuple)) else 0
    def frac(ms, nbytes):
        hbm = nbytes * B / ms / 1e6 / 8000
        valu = (ops * B / ms / 1e6 / 39.3e3) if ops else 0.0
        return max(hbm, valu), ("hbm" if hbm >= valu else "valu")
    rows = []
    acc = torch.zeros(R, dtype=torch.float64, device=dev)
    # leaf-major
    lm = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(lm.data_ptr(), B, L, 1, B, 11, 0, st)
    rlm = torch.empty((R, B), dtype=torch.float64, device=dev).t()
    rows.append(("leaf-major", "eval", timed(lambda: f(rlm, lm)), 8 * (L + R), f.kernel_info()["last_kernel"]))
    rows.append(("leaf-major", "acc", timed(lambda: f.accumulate(lm, None, acc)), 8 * L, f.kernel_info()["last_kernel"]))
    del lm, rlm
    # tile-major
    tm = torch.empty((B // 64, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(tm.data_ptr(), B, L, 1, 64, 64 * L, 11, 0, st)
    rt = torch.empty((B // 64, R, 64), dtype=torch.float64, device=dev)
    rows.append(("tile-major", "eval", timed(lambda: f.eval_tiled(rt, tm, B)), 8 * (L + R), f.kernel_info()["last_kernel"]))
    rows.append(("tile-major", "acc", timed(lambda: f.accumulate_tiled(tm, None, acc, B)), 8 * L, f.kernel_info()["last_kernel"]))
    del tm, rt
    # row-major, contiguous rows
    rm = torch.empty((B, L), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(rm.data_ptr(), B, L, L, 1, 11, 0, st)
    rr = torch.empty((B, R), dtype=torch.float64, device=dev)
    rows.append(("row-major", "eval", timed(lambda: f(rr, rm)), 8 * (L + R), f.kernel_info()["last_kernel"]))
    rows.append(("row-major", "acc", timed(lambda: f.accumulate(rm, None, acc)), 8 * L, f.kernel_info()["last_kernel"]))
    del rm
    # row-major, rows padded by three doubles
    rp = torch.empty((B, L + 3), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(rp.data_ptr(), B, L, L + 3, 1, 11, 0, st)
    rows.append(("row-major padded", "eval", timed(lambda: f(rr, rp[:, :L])), 8 * (L + R), f.kernel_info()["last_kernel"]))
    rows.append(("row-major padded", "acc", timed(lambda: f.accumulate(rp[:, :L], None, acc)), 8 * L, f.kernel_info()["last_kernel"]))
    del rp, rr
    torch.cuda.empty_cache()
    for lay, mode, ms, nb, k in rows:
        fr, bound = frac(ms, nb)
        print(f"{name:24s} L={L:5d} B={B:8d}  {lay:17s} {mode:4s}  {B / ms * 1e3:10.3e} /s  {fr:.3f} of {bound:4s}  [{k}]", flush=True)
