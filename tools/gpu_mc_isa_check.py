"""Monte-Carlo step through ONE kernel of the optimizing back end (route 3) against leaf kernel + evaluator (route 2):
values and throughput (dev tool).  usage: gpu_mc_isa_check.py [workload] [n_sample]"""
def _need_dev_build():
    """This tool steers the library through FDG_* environment variables AFTER it is loaded: only the dev build (make -C feynmandiagram.jl_amd/csrc dev;
    FDG_LIBRARY=.../libfdg_dev.so) reads them then -- the product build snapshots the supported ones once per process and would compare a configuration
    with itself.  Fail loudly instead."""
    import os, sys
    if not os.environ.get("FDG_LIBRARY", "").endswith("libfdg_dev.so"):
        sys.exit(os.path.basename(__file__) + ": needs the dev build (make -C feynmandiagram.jl_amd/csrc dev; export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so): "
                 "the product library reads FDG_* once per process, so the switches this tool flips would be silent no-ops")


_need_dev_build()

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "gv_sigma4_taylor2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
z = dict(np.load(os.path.join(GOLD, ("gv_sigma5" if name.startswith("gv_sigma5") else "gv_sigma4") + "_leafstates.npz")))
if name == "gv_sigma5_taylor2":
    zt = np.load(os.path.join(GOLD, "gv_sigma5_taylor2.npz"))
    for k in ("leaf_type", "tau_in", "tau_out", "loop_index"): z[k] = z[k][zt["leaf_base"]]
    z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
if name == "gv_sigma4_taylor2":
    zt = np.load(os.path.join(GOLD, "gv_sigma4_taylor2.npz"))
    for k in ("leaf_type", "tau_in", "tau_out", "loop_index"): z[k] = z[k][zt["leaf_base"]]
    z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
t = workloads.get(name)
dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"]); n_k = n_loop * dim
kF, beta, lam = 1.919, 3.0, 1.2
X = torch.empty((n_k + n_tau, B), dtype=torch.float64, device=dev)      # ONE component-major matrix: K rows, then T rows
X[:n_k] = torch.rand((n_k, B), dtype=torch.float64, device=dev) * 4 - 2
X[n_k:] = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
dK, dT = X[:n_k], X[n_k:]
dT2 = dT.clone()                                                          # separate allocation: the packing path
w = torch.rand(B, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
def handle(route):
    os.environ["FDG_MC_ROUTE"] = route
    h = fd.compile_table(t, specialize="isa", flags=capi.FDG_SPEC_FAST_MATH if os.environ.get("FAST") else 0).handle
    if len(sys.argv) > 3: h.set_opt_params(*[int(v) for v in sys.argv[3].split(",")])
    h.specialize_fused(tab)
    return h
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
for route in ("split", "isa"):
    h = handle(route)
    root = torch.zeros((t.n_root, B), dtype=torch.float64, device=dev)
    acc = torch.zeros(t.n_root, dtype=torch.float64, device=dev)
    ev = lambda T_=dT: h.mc_eval_device(dK.data_ptr(), 1, B, T_.data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), 1, B, B, st)
    ac = lambda T_=dT: h.mc_accumulate_device(dK.data_ptr(), 1, B, T_.data_ptr(), 1, B, kF, beta, lam, w.data_ptr(), acc.data_ptr(), B, st)
    ev(); torch.cuda.synchronize(); r1 = root.clone()
    acc.zero_(); ac(); torch.cuda.synchronize(); a1 = acc.clone()
    root.zero_(); ev(dT2); torch.cuda.synchronize(); r2 = root.clone()
    res[route] = (r1, a1, r2)
    te, ta, tp = timeit(ev), timeit(ac), timeit(lambda: ac(dT2))
    print(f"{name} route {route}: eval {te:.3f} ms = {B/te*1e3:.3e} samples/s, accumulate {ta:.3f} ms = {B/ta*1e3:.3e} samples/s, "
          f"accumulate with T in its own allocation {tp:.3f} ms = {B/tp*1e3:.3e} samples/s")
(r_s, a_s, r2_s), (r_i, a_i, r2_i) = res["split"], res["isa"]
scale = r_s.abs().amax(dim=1, keepdim=True) + 1e-300
print("roots: max |isa - split| / max|root| =", float(((r_i - r_s).abs() / scale).max()), " packed-path identical to in-place:", bool(torch.equal(r_i, r2_i)))
print("accumulated: rel diff", float(((a_i - a_s).abs() / a_s.abs()).max()), " vs direct sum of roots*w:", float((((r_i * w).sum(dim=1) - a_i).abs() / a_i.abs()).max()))
print("finite:", bool(torch.isfinite(r_i).all()))
