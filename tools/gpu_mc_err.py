"""GPU dev tool: error of the one-kernel Monte-Carlo step's roots against the pure oracle chain (numpy leaves -> oracle graph),
relative to the root's own term scale S_k and to the absolute-value graph A_k."""
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
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
cuda = torch.device("cuda:0")
for name in ("gv_sigma4", "gv_sigma4_taylor2", "gv_sigma5"):
    t, z = workloads.get(name), workloads.leafstates(name)
    L, R = t.n_leaf, t.n_root
    B, dim, n_loop, n_tau = 8011, 3, int(z["basis"].shape[1]), int(z["n_tau"])
    n_k = n_loop * dim
    rng = np.random.default_rng(23)
    K = rng.uniform(-2.0, 2.0, size=(B, n_loop, dim))
    args = (z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
    tab, _keep = capi.make_leaf_tables(*args)
    for route in ("isa", "split"):
        os.environ["FDG_MC_ROUTE"] = route
        g = fd.compile_table(t, specialize="isa")
        g.handle.specialize_fused(tab)
        for kF, beta, lam in ((1.919, 3.0, 1.2), (1.5, 8.0, 0.7)):
            T = rng.uniform(0.0, beta, size=(B, n_tau)); T[:, 0] = 0.0
            h_leaf = oracle.leaf_values(*args[:6], K, T, kF, beta, lam)
            want = oracle.eval_static(t, h_leaf)
            S = np.maximum(1.0, oracle.root_scale(t, h_leaf))
            A = np.maximum(1.0, oracle.abs_graph_scale(t, h_leaf))
            X = torch.from_numpy(np.concatenate([K.reshape(B, n_k).T, T.T], axis=0).copy()).to(cuda)
            root = torch.zeros((B, R), dtype=torch.float64, device=cuda)
            g.handle.mc_eval_device(X.data_ptr(), 1, B, X[n_k:].data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), R, 1, B, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = root.cpu().numpy()
            e = np.abs(got - want)
            print(f"{name} route={route} beta={beta}: max err/S_k {np.nanmax(e / S):.2e}  err/A_k {np.nanmax(e / A):.2e}  A_k/S_k median {np.median(A / S):.1e} max {np.max(A / S):.1e}", flush=True)
