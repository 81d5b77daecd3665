"""Randomised stress of the per-type kernels on the GPU (dev tool): the random DAGs of gpu_fuzz.py (Sum / Prod / Power{2,3}, factors
including -1 and non-powers of two) on Float32 / ComplexF64 / ComplexF32 leaves, row- and column-major, bit for bit against the typed
twin of the oracle; ComplexF64 rows go through the spelled-out graph when it gets a row-major variant.
usage: python tools/gpu_fuzz_typed.py [n_seeds] [first_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd.nodetable import OP_POWER, OP_PROD, OP_SUM, from_program

NP = {"Float32": np.float32, "ComplexF64": np.complex128, "ComplexF32": np.complex64}

def table(seed):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 120)); N = int(rng.choice([5, 40, 300, 900]))
    facs = [1.0, 1.0, 1.0, -1.0, -1.0, 2.0, -0.5, 0.25, 3.0, -7.5, 1e-3, 1.0 / 3.0]
    nodes = []
    for n in range(N):
        nv = L + n; r = rng.random()
        if r < 0.05:
            nodes.append((OP_POWER, int(rng.choice([2, 3])), [(int(rng.integers(0, nv)), float(rng.choice(facs)))])); continue
        op = OP_SUM if r < 0.5 else OP_PROD
        k = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 7, 20]))
        spread = float(rng.choice([3, 20, 200]))
        ch = [(int(nv - 1 - min(nv - 1, int(rng.exponential(spread)))) if rng.random() < 0.7 else int(rng.integers(0, nv)), float(rng.choice(facs))) for _ in range(k)]
        nodes.append((op, 0, ch))
    R = int(rng.integers(1, 6)); roots = [int(rng.integers(0, L + N)) for _ in range(R)]; roots[0] = L + N - 1
    return from_program(L, nodes, roots, f"fuzz_{seed}"), rng

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0; via_view = 0
for seed in range(s0, s0 + n):
    t, rng = table(seed)
    B = int(rng.choice([1, 63, 64, 65, 700, 20_000]))
    f = fd.compile_table(t, specialize="isa", cache_dir="/tmp/fuzz_cache_t")
    for dtype, npdt in NP.items():
        x = (rng.random((B, t.n_leaf)) * 1.6 - 0.5)
        if dtype.startswith("Complex"): x = x + 1j * (rng.random(x.shape) * 1.6 - 0.9)
        x = x.astype(npdt)
        with np.errstate(all="ignore"):
            want = oracle.eval_static_typed(t, x, dtype)
        for layout in ("rows", "columns"):
            leaf = torch.from_numpy(x).to(dev) if layout == "rows" else torch.from_numpy(np.ascontiguousarray(x.T)).to(dev).t()
            got = f(None, leaf); torch.cuda.synchronize()
            got = np.ascontiguousarray(got.cpu().numpy())
            via_view += "ComplexF64 rows" in f.last_typed_kernel
            gv, wv = got.view(NP[dtype]).view(np.float32 if "32" in dtype else np.float64), np.ascontiguousarray(want).view(np.float32 if "32" in dtype else np.float64)
            nanm = np.isnan(wv)
            ui = np.uint32 if "32" in dtype else np.uint64       # bit for bit outside the NaNs: the signs of zeros included
            if not (np.array_equal(np.isnan(gv), nanm) and np.array_equal(gv[~nanm].view(ui), wv[~nanm].view(ui))):
                print("MISMATCH seed", seed, dtype, layout, "B", B, "L", t.n_leaf, "N", t.n_node, f.last_typed_kernel); bad += 1
    if (seed - s0) % 10 == 9: print("..", seed - s0 + 1, "seeds, bad =", bad, " complex-row launches through the spelled-out graph:", via_view, flush=True)
print("done:", n, "seeds, bad =", bad)
sys.exit(1 if bad else 0)
