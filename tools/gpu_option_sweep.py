"""GPU dev tool (round 5): one workload (tile-major batch; SWEEP_LAYOUT=rm: compile_Python's row-major [B, L] / [B, R]; SWEEP_LAYOUT=lm: a Julia
column-major pair) under sets of handle OPTIONS (fdg_graph_set_option; works with the product build):
rate, kernel, bitwise check against the first set.   usage: gpu_option_sweep.py workload B "K1=V1,K2=V2" "K3=V3" ...   ("-" = no option)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
sets = sys.argv[3:] or ["-"]
t = workloads.get(name); L, R = t.n_leaf, t.n_root
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
LAYOUT = os.environ.get("SWEEP_LAYOUT", "tm")
if LAYOUT == "rm":
    leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(leaf.data_ptr(), B, L, L, 1, 1234, 0, st)
    root = torch.zeros((B, R), dtype=torch.float64, device=dev)
elif LAYOUT == "lm":
    leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(leaf.data_ptr(), B, L, 1, B, 1234, 0, st)
    root = torch.zeros((R, B), dtype=torch.float64, device=dev).t()
else:
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)      # ONE root array for every set: the rate depends on the (leaf pages, root pages) pair
ref = None
def run(f):
    if LAYOUT in ("rm", "lm"): f(root, leaf)
    else: f.eval_tiled(root, leaf, B)
for spec in sets:
    opts = {} if spec == "-" else dict(kv.split("=") for kv in spec.split(","))
    opts["FDG_IGNORE_TUNED"] = opts.get("FDG_IGNORE_TUNED", None)
    opts = {k: v for k, v in opts.items() if v is not None}
    t0 = time.time()
    try:
        f = fd.compile_table(t, specialize="isa", options=opts, cache_dir="/tmp/sweep_cache")
    except capi.FdgError as e:
        print(f"{name} [{spec}] specialize failed: {str(e)[:120]}", flush=True); continue
    tc = time.time() - t0
    root.zero_()
    for _ in range(40): run(f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): run(f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    if ref is None: ref = root.clone()
    ki = f.kernel_info()
    print(f"{name} [{spec}] {ki['last_kernel']:20s} {ms:7.3f} ms {B / ms / 1e3:9.1f} Mevals/s frac_hbm {8 * (L + R) * B / ms / 1e6 / 8000:.3f}  same bits: {bool(torch.equal(root, ref))}  (compile {tc:.1f} s)", flush=True)
