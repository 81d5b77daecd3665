"""GPU dev tool: one seed of tests/test_random_graphs.py's fuzz with one option set, verbose."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi
import test_random_graphs as T
seed = int(sys.argv[1]); which = int(sys.argv[2])
dev = torch.device("cuda:0")
t, rng = T.fuzz_table(seed)
B = int(rng.choice([1, 63, 64, 65, 700, 140_000]))
h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 2 - 0.7
want = oracle.eval_static(t, h_leaf, np.full((B, t.n_root), 9.0))
opts = [dict(n_reg=int(rng.integers(6, 40)), n_lds=int(rng.integers(1, 30)), vn_window=int(rng.choice([1, 20, 200, 1000]))),
        dict(n_reg=int(rng.integers(30, 120)), n_lds=int(rng.integers(1, 80)), n_acc=int(rng.integers(1, 124))), None]
opt = opts[which]
print("seed", seed, "L", t.n_leaf, "N", t.n_node, "R", t.n_root, "B", B, opt, {k: v for k, v in os.environ.items() if k.startswith("FDG_")})
for rep in range(3):
    f = fd.compile_table(t, specialize="isa", opt=opt, cache_dir="/tmp/fz")
    for layout in ("leaf_major", "sample_major"):
        leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(dev).t() if layout == "leaf_major" else torch.from_numpy(h_leaf).to(dev)
        root = torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=dev)
        f(root, leaf); torch.cuda.synchronize()
        got = root.cpu().numpy()
        bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
        print(rep, layout, "bad", int(bad.sum()), "rows", np.unique(np.nonzero(bad)[0])[:10], "cols", np.unique(np.nonzero(bad)[1]))
