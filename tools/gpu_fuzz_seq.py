"""GPU dev tool: the fuzz test's sequence of seeds in one process, with mismatch details."""
import os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi
import test_random_graphs as T
first, last = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
os.makedirs("/tmp/fzs", mode=0o700, exist_ok=True)
for seed in range(first, last + 1):
    t, rng = T.fuzz_table(seed)
    B = int(rng.choice([1, 63, 64, 65, 700, 140_000]))
    h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 2 - 0.7
    want = oracle.eval_static(t, h_leaf, np.full((B, t.n_root), 9.0))
    opts = [dict(n_reg=int(rng.integers(6, 40)), n_lds=int(rng.integers(1, 30)), vn_window=int(rng.choice([1, 20, 200, 1000]))),
            dict(n_reg=int(rng.integers(30, 120)), n_lds=int(rng.integers(1, 80)), n_acc=int(rng.integers(1, 124)))]
    for k in ("FDG_ISA_W2", "FDG_REMAT_WINDOW", "FDG_ISA_COOP"): capi.set_default_option(k, None)
    if seed % 4 == 0: capi.set_default_option("FDG_ISA_W2", "1"); opts.append(None)
    if seed % 3 == 0: capi.set_default_option("FDG_REMAT_WINDOW", str(int(rng.choice([8, 60, 400]))))
    if seed % 2 == 1: capi.set_default_option("FDG_ISA_COOP", "1")
    for opt in opts:
        f = fd.compile_table(t, specialize="isa", opt=opt, cache_dir="/tmp/fzs", flags=capi.FDG_SPEC_KEEP_SOURCE)
        for layout in ("leaf_major", "sample_major"):
            leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(dev).t() if layout == "leaf_major" else torch.from_numpy(h_leaf).to(dev)
            root = (torch.full((t.n_root, B), 9.0, dtype=torch.float64, device=dev).t() if layout == "leaf_major" and seed % 2 == 0
                    else torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=dev))
            f(root, leaf); torch.cuda.synchronize()
            got = root.cpu().numpy()
            bad = ~((got == want) | (np.isnan(got) & np.isnan(want))) | ((np.signbit(got) != np.signbit(want)) & ~np.isnan(want))
            if bad.any():
                rows = np.unique(np.nonzero(bad)[0])
                print("MISMATCH seed", seed, opt, layout, "B", B, "nbad", int(bad.sum()), "rows", rows[:8], "...", rows[-3:], "rows%64", np.unique(rows % 64)[:20], "tiles", np.unique(rows // 64)[:10], "cols", np.unique(np.nonzero(bad)[1]), "got", got[bad][:4], "want", want[bad][:4], flush=True)
        w = torch.rand(B, dtype=torch.float64, device=dev)
        acc = f.accumulate(leaf, w); torch.cuda.synchronize()
    print("seed", seed, "done", B, t.n_node, flush=True)
