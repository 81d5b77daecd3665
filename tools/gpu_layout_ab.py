"""GPU dev tool: leaf-major matrix vs tile-major batch (fdg_eval_device_tiled), same Philox values, same process, per workload.
usage: gpu_layout_ab.py workload[:B] ...      (B defaults to ~2.4 GB of leaves)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=10, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for spec in sys.argv[1:]:
    name, _, b = spec.partition(":")
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    B = int(b) if b else max(1 << 14, min(8_000_000, int(2.4e9 / (8 * L)))) // 64 * 64
    T = B // 64
    f = fd.compile_table(t, specialize="isa")
    h = f.handle
    leaf_c = torch.empty((L, B), dtype=torch.float64, device=dev); root_c = torch.zeros((R, B), dtype=torch.float64, device=dev)
    leaf_t = torch.empty((T, L, 64), dtype=torch.float64, device=dev); root_t = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(leaf_c.data_ptr(), B, L, 1, B, 1234, 0, st)
    capi.fill_uniform_device_tiled(leaf_t.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    lm = timed(lambda: h.eval_device(leaf_c.data_ptr(), 1, B, root_c.data_ptr(), 1, B, B, st)); k_lm = f.kernel_info()["last_kernel"]
    tm = timed(lambda: h.eval_device_tiled(leaf_t.data_ptr(), 1, 64, 64 * L, root_t.data_ptr(), 1, 64, 64 * R, B, st)); k_tm = f.kernel_info()["last_kernel"]
    same = all(bool(torch.equal(root_t[:, k, :].reshape(-1), root_c[k])) for k in range(R))
    acc = torch.zeros(R, dtype=torch.float64, device=dev)
    alm = timed(lambda: h.accumulate_device(leaf_c.data_ptr(), 1, B, 0, acc.data_ptr(), B, st))
    atm = timed(lambda: h.accumulate_device_tiled(leaf_t.data_ptr(), 1, 64, 64 * L, 0, acc.data_ptr(), B, st))
    print(f"{name:24s} B={B:9d} L={L:5d}  eval: leaf-major {B / lm * 1e3:.3e}/s ({k_lm})  tile-major {B / tm * 1e3:.3e}/s ({k_tm})  ratio {lm / tm:.3f} bitwise equal {same}"
          f"  |  accumulate: {B / alm * 1e3:.3e}/s  {B / atm * 1e3:.3e}/s  ratio {alm / atm:.3f}", flush=True)
    del leaf_c, leaf_t, root_c, root_t, f
    torch.cuda.empty_cache()
