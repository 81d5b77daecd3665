"""GPU dev tool (round 6): evaluation AND fused accumulation of one workload (tile-major batch) under explicit register budgets, A B A B in one process.
usage: gpu_cfg_ab_acc.py workload B "n_reg=120,n_lds=40,vn_window=200" "n_reg=120,n_lds=80,n_acc=124,vn_window=400" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
t = workloads.get(name); L, R = t.n_leaf, t.n_root
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
w = torch.rand(B, dtype=torch.float64, device=dev)
acc = torch.zeros(R, dtype=torch.float64, device=dev)
fs = []
for spec in sys.argv[3:]:
    opt = {k: int(v) for k, v in (x.split("=") for x in spec.split(","))}
    fs.append((spec, fd.compile_table(t, specialize="isa", cache_dir="/tmp/fdg-sweep-cache", opt=opt)))
def timed(fn, n=40):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    for spec, f in fs:
        me = timed(lambda: f.eval_tiled(root, leaf, B))
        ma = timed(lambda: f.accumulate_tiled(leaf, w, acc, B))
        print(f"{name} [{spec}] eval {me:.3f} ms {B / me / 1e3:8.1f} M/s ({f.kernel_info()['last_kernel']}) | accumulate {ma:.3f} ms {B / ma / 1e3:8.1f} M/s", flush=True)
