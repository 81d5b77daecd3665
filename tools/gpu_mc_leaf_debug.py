"""Leaf formulas of the fused ISA step one by one: a graph whose roots ARE its leaves, against fdg_leaf_eval_device (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads
from feynmandiagram_jl_amd.nodetable import NodeTable
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = dict(np.load(os.path.join(GOLD, "gv_sigma4_leafstates.npz")))
if len(sys.argv) > 1 and sys.argv[1] == "taylor2":
    zt = np.load(os.path.join(GOLD, "gv_sigma4_taylor2.npz"))
    for k in ("leaf_type", "tau_in", "tau_out", "loop_index"): z[k] = z[k][zt["leaf_base"]]
    z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
if len(sys.argv) > 1 and sys.argv[1] == "orders":     # green_derive orders 0..5 on every fermionic leaf
    z["leaf_order"] = np.where(z["leaf_type"] == 1, np.arange(len(z["leaf_type"])) % 6, np.arange(len(z["leaf_type"])) % 4).astype(np.int32)
L = len(z["leaf_type"])
sub = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(L))
t = NodeTable(L, np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0), np.array(sub, np.uint32), "leaves")
dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"]); n_k = n_loop * dim
kF, beta, lam = 1.919, float(os.environ.get("BETA", 3.0)), 1.2
KMAX = float(os.environ.get("KMAX", 2.0))
B = 64 * 300 + 17
X = torch.empty((n_k + n_tau, B), dtype=torch.float64, device=dev)
X[:n_k] = (torch.rand((n_k, B), dtype=torch.float64, device=dev) * 2 - 1) * KMAX
X[n_k:] = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
X[n_k + 1, :5] = X[n_k, :5]     # tau == 0
st = torch.cuda.current_stream().cuda_stream
leaf = torch.zeros((L, B), dtype=torch.float64, device=dev)
capi.leaf_eval_device(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau, kF, beta, lam,
                      X.data_ptr(), 1, B, X[n_k:].data_ptr(), 1, B, leaf.data_ptr(), 1, B, B, st)
tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
os.environ["FDG_MC_ROUTE"] = "isa"
h = fd.compile_table(t, specialize="isa").handle
h.specialize_fused(tab, flags=capi.FDG_SPEC_KEEP_SOURCE)
root = torch.zeros((len(sub), B), dtype=torch.float64, device=dev)
h.mc_eval_device(X.data_ptr(), 1, B, X[n_k:].data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), 1, B, B, st)
torch.cuda.synchronize()
r1 = root.clone(); root.zero_()
h.mc_eval_device(X.data_ptr(), 1, B, X[n_k:].data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), 1, B, B, st)
torch.cuda.synchronize()
print("deterministic:", bool(torch.equal(r1, root)))
want = leaf[sub]
err = ((root - want).abs() / (want.abs() + 1e-290)).amax(dim=1).cpu().numpy()     # (results below 1e-290 count as zero)
for j, i in enumerate(sub):
    if err[j] > 1e-12 or len(sub) <= 8:
        print(f"leaf {i}: type {z['leaf_type'][i]} order {z['leaf_order'][i]} loop {z['loop_index'][i]} tau {z['tau_in'][i]}->{z['tau_out'][i]}  max rel err {err[j]:.3e}  e.g. got {float(root[j,7]):.6e} want {float(want[j,7]):.6e}")
print("leaves checked", len(sub), "worst", err.max(), "bad", int((err > 1e-12).sum()))
if len(sub) <= 8:
    j = 0; i = sub[0]
    e = ((root[j] - want[j]).abs() / (want[j].abs() + 1e-300))
    bad = torch.argsort(e, descending=True)[:6].cpu().numpy()
    Kh = X[:n_k].cpu().numpy(); Th = X[n_k:].cpu().numpy()
    bv = z["basis"][z["loop_index"][i] - 1]
    for b in bad:
        q = (Kh[:, b].reshape(n_loop, dim) * bv[:, None]).sum(axis=0); q2 = (q * q).sum(); w_ = q2 - kF * kF
        tau = Th[z["tau_out"][i] - 1, b] - Th[z["tau_in"][i] - 1, b]
        print(f"sample {b}: q2 {q2:.4f} w {w_:.4f} -|w|beta {-abs(w_)*beta:.3f} tau {tau:.4f} got {float(root[j,b]):.6e} want {float(want[j,b]):.6e}")
if os.environ.get("FDG_MC_DEBUG_STAGE") and len(sub) <= 8:
    i = sub[0]; bv = z["basis"][z["loop_index"][i] - 1]
    Kh = X[:n_k].cpu().numpy(); Th = X[n_k:].cpu().numpy(); got = root[0].cpu().numpy()
    q = (Kh.reshape(n_loop, dim, B) * bv[:, None, None]).sum(axis=0); w_ = (q * q).sum(axis=0) - kF * kF
    tau = Th[z["tau_out"][i] - 1] - Th[z["tau_in"][i] - 1]; tau = np.where(tau == 0, -1e-10, tau)
    a_ = np.where(w_ > 0, np.where(tau > 0, -tau, -(tau + beta)), np.where(tau > 0, beta - tau, -tau))
    ref = {"w": w_, "g": 1 / (1 + np.exp(-np.abs(w_) * beta)), "tau": tau, "a": a_, "A": np.exp(w_ * a_), "wa": w_ * a_,
           "u": np.where(tau > 0, tau, tau + beta), "v": np.where(tau > 0, tau - beta, tau)}[os.environ["FDG_MC_DEBUG_STAGE"]]
    e = np.abs(got - ref) / (np.abs(ref) + 1e-300)
    bad = np.argsort(-e)[:5]
    print("stage", os.environ["FDG_MC_DEBUG_STAGE"], "worst rel err", e.max())
    for b in bad: print(f"  sample {b}: w {w_[b]:.4f} tau {tau[b]:.4f} got {got[b]:.10e} ref {ref[b]:.10e}")
