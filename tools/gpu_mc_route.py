"""GPU dev tool: the whole Monte-Carlo step (leaves from K, T + graph) of a workload through the routes of fdg_mc_eval_device.
python tools/gpu_mc_route.py WORKLOAD B [route ...]   routes: default isa leafkernel"""
def _need_dev_build():
    """This tool steers the library through FDG_* environment variables AFTER it is loaded: only the dev build (make -C feynmandiagram.jl_amd/csrc dev;
    FDG_LIBRARY=.../libfdg_dev.so) reads them then -- the product build snapshots the supported ones once per process and would compare a configuration
    with itself.  Fail loudly instead."""
    import os, sys
    if not os.environ.get("FDG_LIBRARY", "").endswith("libfdg_dev.so"):
        sys.exit(os.path.basename(__file__) + ": needs the dev build (make -C feynmandiagram.jl_amd/csrc dev; export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so): "
                 "the product library reads FDG_* once per process, so the switches this tool flips would be silent no-ops")


_need_dev_build()

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name, B = sys.argv[1], int(sys.argv[2])
t, z = workloads.get(name), workloads.leafstates(name)
dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"])
kF, beta, lam = 1.919, 3.0, 1.2
dK = torch.rand((n_loop * dim, B), dtype=torch.float64, device=dev) * 4 - 2
dT = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
root = torch.zeros((t.n_root, B), dtype=torch.float64, device=dev).t()
tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau, kF, beta, lam)
st = torch.cuda.current_stream().cuda_stream
ref = None
for route in sys.argv[3:] or ["default"]:
    if route == "default": os.environ.pop("FDG_MC_ROUTE", None)
    else: os.environ["FDG_MC_ROUTE"] = route
    try:
        t0 = time.time()
        h = fd.compile_table(t, specialize="isa", cache_dir="/tmp/fdg-sweep-cache").handle
        h.specialize_fused(tab)
        tc = time.time() - t0
        run = lambda: h.mc_eval_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), 1, B, B, st)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        got = root[:1000].cpu().numpy().copy()
        dev_ = 0.0 if ref is None else float(np.max(np.abs(got - ref) / np.maximum(1e-300, np.abs(ref))))
        if ref is None: ref = got
        print(f"{name} route {route}: {ms:.3f} ms {B / ms * 1e3:.3e} samples/s  specialise {tc:.1f} s  max rel dev vs first route {dev_:.1e}", flush=True)
    except Exception as e:
        print(f"{name} route {route}: FAILED {e}", flush=True)
