"""Randomised stress of the optimizing back end on the GPU (dev tool): random DAGs of varied size through
random register / LDS / AGPR budgets, eval in three layouts (leaf-major, row-major, tile-major) + fused accumulate, 1 ... 130 roots
(VGPR and AGPR accumulators, the root scratch for row-major root matrices), against the oracle (bit-exact).
usage: python tools/gpu_fuzz.py [n_seeds] [first_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi
from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT, OP_POWER, OP_PROD, OP_SUM, from_program

def table(seed):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 200)); N = int(rng.choice([5, 40, 300, 1500]))
    facs = [1.0, 1.0, 1.0, -1.0, -1.0, 2.0, -0.5, 0.25, 3.0, -7.5, 1e-3, 1.0 / 3.0]
    nodes = []
    for n in range(N):
        nv = L + n; r = rng.random()
        if r < 0.05:
            nodes.append((OP_POWER, int(rng.choice([2, 3, 2, 3, 4, 5, -1, -2, -3])), [(int(rng.integers(0, nv)), float(rng.choice(facs)))])); continue
        op = OP_SUM if r < 0.45 else OP_PROD
        k = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 7, 30]))
        spread = float(rng.choice([3, 20, 200]))
        ch = [(int(nv - 1 - min(nv - 1, int(rng.exponential(spread)))) if rng.random() < 0.7 else int(rng.integers(0, nv)), float(rng.choice(facs))) for _ in range(k)]
        nodes.append((op, 0, ch))
    R = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 7, 17, 30, 45, 90, 130])); roots = [int(rng.integers(0, L + N)) for _ in range(R)]; roots[0] = L + N - 1
    return from_program(L, nodes, roots, f"fuzz_{seed}"), rng

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for seed in range(s0, s0 + n):
    t, rng = table(seed)
    B = int(rng.choice([1, 63, 64, 65, 700, 140_000]))
    h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 2 - 0.7
    want = oracle.eval_static(t, h_leaf, np.full((B, t.n_root), 9.0))
    opts = [None, dict(n_reg=int(rng.integers(6, 40)), n_lds=int(rng.integers(1, 30)), vn_window=int(rng.choice([1, 20, 200, 1000]))),
            dict(n_reg=int(rng.integers(30, 120)), n_lds=int(rng.integers(1, 80)), n_acc=int(rng.integers(1, 124)))]
    for opt in opts:
        try:
            f = fd.compile_table(t, specialize="isa", opt=opt, cache_dir="/tmp/fuzz_cache")
        except capi.FdgError as e:
            print("seed", seed, opt, "specialize:", e); bad += 1; continue
        for layout in ("leaf_major", "sample_major"):
            leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(dev).t() if layout == "leaf_major" else torch.from_numpy(h_leaf).to(dev)
            root = torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=dev)
            f(root, leaf); torch.cuda.synchronize()
            got = root.cpu().numpy()
            nanm = np.isnan(want)
            if not (np.array_equal(np.isnan(got), nanm) and np.array_equal(got[~nanm], want[~nanm])):
                print("MISMATCH seed", seed, opt, layout, "B", B, "L", t.n_leaf, "N", t.n_node); bad += 1
        # tile-major batch: evaluation and accumulation
        T = (B + 63) // 64
        full = np.full((T * 64, t.n_leaf), np.nan); full[:B] = h_leaf
        tl = torch.from_numpy(np.ascontiguousarray(full.reshape(T, 64, t.n_leaf).transpose(0, 2, 1))).to(dev)
        rt = torch.full((T, t.n_root, 64), 9.0, dtype=torch.float64, device=dev)
        f.eval_tiled(rt, tl, B); torch.cuda.synchronize()
        got = rt.cpu().numpy().transpose(0, 2, 1).reshape(T * 64, t.n_root)
        nanm = np.isnan(want)
        if not (np.array_equal(np.isnan(got[:B]), nanm) and np.array_equal(got[:B][~nanm], want[~nanm]) and (got[B:] == 9.0).all()):
            print("MISMATCH seed", seed, opt, "tile_major", "B", B, "L", t.n_leaf, "N", t.n_node, "R", t.n_root); bad += 1
        w = torch.rand(B, dtype=torch.float64, device=dev)
        acc_t = f.accumulate_tiled(tl, w, None, B); torch.cuda.synchronize()
        acc = f.accumulate(leaf, w); torch.cuda.synchronize()
        lm = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(dev).t()
        acc_l = f.accumulate(lm, w); torch.cuda.synchronize()
        live = t.root_slot != FDG_NO_ROOT
        wr = np.where(np.isnan(want), 0.0, want) * w.cpu().numpy()[:, None]
        for which, a in (("row-major", acc), ("tile-major", acc_t), ("leaf-major", acc_l)):
            ok = np.abs(a.cpu().numpy() - wr.sum(0))[live] <= 1e-12 * np.maximum(1.0, np.abs(wr).sum(0))[live]
            if np.isfinite(want).all() and not ok.all():      # (overflowing graphs make the sum inf/nan on both sides)
                print("ACC MISMATCH seed", seed, opt, which, "B", B, "R", t.n_root); bad += 1
    if (seed - s0) % 20 == 19: print("..", seed - s0 + 1, "seeds, bad =", bad, flush=True)
print("done:", n, "seeds, bad =", bad)
sys.exit(1 if bad else 0)
