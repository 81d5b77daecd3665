"""GPU dev tool: one workload's ISA kernel under explicit register budgets (fdg_graph_set_opt_params) at a given batch.
python tools/gpu_cfg_sweep.py WORKLOAD B "n_reg=56,n_lds=20" "n_reg=28,n_lds=1" ...   ("-" = the library's choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
t = workloads.get(name)
st = t.stats()
L, R = t.n_leaf, t.n_root
TILED = os.environ.get("SWEEP_LAYOUT") == "tile_major"
if TILED:
    B = (B + 63) // 64 * 64
    leaf = torch.empty((B // 64, L, 64), dtype=torch.float64, device=dev)
    root = torch.empty((B // 64, R, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 11, 0, torch.cuda.current_stream().cuda_stream)
else:
    leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
    capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 11, 0, torch.cuda.current_stream().cuda_stream)
nchk = 4099
rows = lambda x, n: (x[:(n + 63) // 64].permute(0, 2, 1).reshape(-1, x.shape[1])[:n] if TILED else x[:n]).cpu().numpy()
run = lambda f: f.eval_tiled(root, leaf, B) if TILED else f(root, leaf)
want = oracle.eval_static(t, rows(leaf, nchk), np.zeros((nchk, R)))
for setting in sys.argv[3:]:
    opt = None if setting == "-" else {k: int(v) for k, v in (x.split("=") for x in setting.split(","))}
    try:
        f = fd.compile_table(t, specialize="isa", cache_dir="/tmp/fdg-sweep-cache", opt=opt)
        root.zero_()
        run(f); torch.cuda.synchronize()
        ok = np.array_equal(rows(root, nchk), want)
        for _ in range(20): run(f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = int(os.environ.get("SWEEP_N", 20))
        e0.record()
        for _ in range(n): run(f)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        i = f.info()
        print(f"{name} [{setting}] {'exact' if ok else 'MISMATCH'} B={B} {ms:.3f} ms {B / ms * 1e3:.3e} evals/s  alg {B / ms * 1e3 * st['bytes_alg'] / 1e9:.0f} GB/s "
              f"vgpr={i['spec_vgpr']} lds={i['spec_lds_bytes']}", flush=True)
        del f
    except Exception as e:
        print(f"{name} [{setting}] FAILED: {e}", flush=True)
