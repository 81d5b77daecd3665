"""Soak of the one-kernel Monte-Carlo step: many batches of fresh random inputs, each compared with leaf kernel + evaluator
(tolerance-level differences only: own exp vs ocml's) and with a second run of itself (bits) -- a stale read from a missed
hardware hazard would show as a large deviation in a few samples (dev tool).  usage: gpu_mc_soak.py [workload] [batches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "gv_sigma4"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
z = dict(np.load(os.path.join(GOLD, ("gv_sigma5" if name.startswith("gv_sigma5") else "gv_sigma4") + "_leafstates.npz")))
if name.endswith("_taylor2"):
    zt = np.load(os.path.join(GOLD, name + ".npz"))
    for k in ("leaf_type", "tau_in", "tau_out", "loop_index"): z[k] = z[k][zt["leaf_base"]]
    z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
t = workloads.get(name)
B, dim, n_loop, n_tau = 2_000_003, 3, int(z["basis"].shape[1]), int(z["n_tau"]); n_k = n_loop * dim
tab, _k = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
hs = {}
for route in ("split", "isa"):
    os.environ["FDG_MC_ROUTE"] = route
    hs[route] = fd.compile_table(t, specialize="isa").handle; hs[route].specialize_fused(tab)
st = torch.cuda.current_stream().cuda_stream
worst = 0.0
for it in range(nb):
    kF, beta, lam = 1.0 + 0.05 * (it % 7), [0.5, 3.0, 10.0, 40.0][it % 4], 0.5 + 0.1 * (it % 5)
    K = (torch.rand((n_k, B), dtype=torch.float64, device=dev) * 2 - 1) * [1.0, 2.0, 4.0][it % 3]
    T = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
    out = {}
    for route in ("split", "isa", "isa2"):
        r = torch.zeros((t.n_root, B), dtype=torch.float64, device=dev)
        hs[route[:3] if route != "split" else route].mc_eval_device(K.data_ptr(), 1, B, T.data_ptr(), 1, B, kF, beta, lam, r.data_ptr(), 1, B, B, st)
        out[route] = r
    torch.cuda.synchronize()
    assert torch.equal(out["isa"], out["isa2"]), f"batch {it}: two runs of the same kernel differ"
    assert bool(torch.isfinite(out["isa"]).all()), f"batch {it}: non-finite root"
    scale = out["split"].abs().amax(dim=1, keepdim=True)
    # samplewise: difference against the larger of |root| and 1e-6 of the batch's largest root (cancelling samples amplify a last-bit leaf difference)
    dev_ = ((out["isa"] - out["split"]).abs() / torch.maximum(out["split"].abs(), 1e-6 * scale)).max().item()
    worst = max(worst, dev_)
    assert dev_ < 1e-6, f"batch {it} (kF {kF} beta {beta} lambda {lam}): route isa deviates from leaf kernel + evaluator by {dev_:.3e}"
print(f"{name}: {nb} batches x {B} samples, both routes agree (worst samplewise deviation {worst:.2e}), the one-kernel route is deterministic")
