"""GPU dev tool: one fuzz seed through all three back ends vs the oracle, with zero-sign check."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
import test_random_graphs as T
seed = int(sys.argv[1])
dev = torch.device("cuda:0")
t, rng = T.fuzz_table(seed)
B = 20000
h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 2 - 0.7
want = oracle.eval_static(t, h_leaf)
wn = oracle.eval_static_numpy(t, h_leaf)
print("oracle C vs numpy twin same:", T.same(wn, want))
for spec in ("isa", True, False):
    f = fd.compile_table(t, specialize=spec, cache_dir="/tmp/fzb")
    leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(dev).t()
    root = torch.zeros((B, t.n_root), dtype=torch.float64, device=dev)
    f(root, leaf); torch.cuda.synchronize()
    got = root.cpu().numpy()
    sg = (np.signbit(got) != np.signbit(want)) & ~np.isnan(want)
    print(spec, "same", T.same(got, want), "sign mismatches per root", sg.sum(0), "zeros in want per root", (want == 0).sum(0))
