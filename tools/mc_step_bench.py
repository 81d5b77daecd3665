"""Whole integrand step on device: (K, T) -> leaves -> graph -> weighted accumulation (dev tool)."""
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
z = dict(np.load(os.path.join(GOLD, "gv_sigma4_leafstates.npz")))
name = sys.argv[1] if len(sys.argv) > 1 else "gv_sigma4"
t = workloads.get(name); L = t.n_leaf
if name == "gv_sigma4_taylor2":
    # leaves of the Taylor-expanded graph = (leaf of the 4-loop graph, derivative order in the coupling)
    zt = np.load(os.path.join(GOLD, "gv_sigma4_taylor2.npz"))
    base, dord = zt["leaf_base"], zt["leaf_dorder"]
    for k in ("leaf_type", "tau_in", "tau_out", "loop_index"):
        z[k] = z[k][base]
    z["leaf_order"] = np.where(z["leaf_type"] == 2, dord, 0).astype(np.int32)
    assert np.all(dord[z["leaf_type"] == 1] == 0)
B, dim, n_loop, n_tau = (1 << 22) if name == "gv_sigma4" else 2_000_000, 3, int(z["basis"].shape[1]), int(z["n_tau"])
kF, beta, lam = 1.919, 3.0, 1.2
dK = (torch.rand((n_loop * dim, B), dtype=torch.float64, device=dev) * 4 - 2)
dT = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
leaf = torch.zeros((L, B), dtype=torch.float64, device=dev).t()
w = torch.rand(B, dtype=torch.float64, device=dev)
f = fd.compile_table(t, specialize="isa")
st = torch.cuda.current_stream().cuda_stream
def leaves():
    capi.leaf_eval_device(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau, kF, beta, lam,
                          dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, leaf.data_ptr(), leaf.stride(0), leaf.stride(1), B, st)
acc = torch.zeros(t.n_root, dtype=torch.float64, device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
os.environ["FDG_LEAF_GENERIC"] = "1"
tg = timeit(leaves)
del os.environ["FDG_LEAF_GENERIC"]
print(f"table-driven leaf kernel: {tg:.3f} ms")
tl = timeit(leaves); te = timeit(lambda: f.accumulate(leaf, w, acc))
def both(): leaves(); f.accumulate(leaf, w, acc)
tb = timeit(both)
print(f"{name} B={B}: leaves {tl:.3f} ms ({B/tl*1e3:.3e}/s, {B*L*8/tl/1e6:.0f} GB/s written), eval+accumulate {te:.3f} ms ({B/te*1e3:.3e}/s), whole step {tb:.3f} ms = {B/tb*1e3:.3e} samples/s")

# fused kernel (leaves in registers)
tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
hf = fd.compile_table(t, specialize="isa").handle; hf.specialize_fused(tab)
tf = timeit(lambda: hf.mc_accumulate_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, w.data_ptr(), acc.data_ptr(), B, st))
print(f"fdg_mc_accumulate_device, route chosen by the library: {tf:.3f} ms = {B/tf*1e3:.3e} samples/s")
# (all three routes of fdg_graph_specialize_fused side by side, with value checks: tools/gpu_mc_isa_check.py)
