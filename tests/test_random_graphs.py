"""Randomised structure tests: small random DAGs (Sum/Prod/Power{2,3}, shared nodes, duplicate children,
factors incl. +-1 and 0-free, leaves as roots, interior roots, dead code) through
 * the allocator/scheduler replay on the CPU (every register budget), and
 * (GPU) all three back ends,
against the oracle, bit for bit."""
import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi
from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT, OP_POWER, OP_PROD, OP_SUM, from_program
from test_next_rows import replay


def same(got, want):
    """bitwise equality up to NaN payloads (NaNs must sit at the same places; zeros keep their sign)"""
    n = np.isnan(want)
    return (np.array_equal(np.isnan(got), n) and np.array_equal(got[~n], want[~n])
            and np.array_equal(np.signbit(got[~n]), np.signbit(want[~n])))


def random_table(seed: int):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 12))
    N = int(rng.integers(1, 60))
    facs = [1.0, 1.0, 1.0, -1.0, -1.0, 2.0, -0.5, 0.25, 3.0, -7.5, 1e-3, 1.0 / 3.0]
    nodes = []
    for n in range(N):
        nv = L + n
        r = rng.random()
        if r < 0.12:
            nodes.append((OP_POWER, int(rng.choice([2, 3])), [(int(rng.integers(0, nv)), float(rng.choice(facs)))]))
            continue
        op = OP_SUM if r < 0.5 else OP_PROD
        k = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 7, 19]))
        ch = []
        for _ in range(k):
            # bias towards recent values, allow duplicates
            c = int(nv - 1 - min(nv - 1, int(rng.exponential(6)))) if rng.random() < 0.7 else int(rng.integers(0, nv))
            ch.append((c, float(rng.choice(facs))))
        nodes.append((op, 0, ch))
    R = int(rng.integers(1, 6))
    roots = [int(rng.integers(0, L + N)) for _ in range(R)]
    roots[0] = L + N - 1
    if R > 2 and rng.random() < 0.3:
        roots[2] = FDG_NO_ROOT
    return from_program(L, nodes, roots, f"random_{seed}")


SEEDS = list(range(40))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_graph_allocated_program_replays(libfdg, seed):
    t = random_table(seed)
    h = capi.GraphHandle(t)
    leaf = oracle.philox_uniform(7, t.n_leaf, seed) * 4 - 2
    want = oracle.eval_static(t, leaf)
    rng = np.random.default_rng(seed)
    for budget in (dict(), dict(n_reg=int(rng.integers(4, 12)), n_lds=int(rng.integers(1, 4)), n_acc=int(rng.integers(1, 4)),
                                lookahead_leaf=int(rng.integers(1, 30)), vn_window=int(rng.choice([1, 5, 200])))):
        ops, nr, nl, nm = h.opt_program(**budget)
        got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
        live = t.root_slot != FDG_NO_ROOT
        assert same(got[:, live], want[:, live]), (seed, budget)


@pytest.mark.gpu
def test_random_graphs_on_device(libfdg, cuda):
    import torch
    for seed in SEEDS:
        t = random_table(seed)
        B = int(np.random.default_rng(seed).choice([1, 65, 300]))
        h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 4 - 2
        want = oracle.eval_static(t, h_leaf, np.full((B, t.n_root), 9.0))
        for spec in ("isa", True, False):
            f = fd.compile_table(t, specialize=spec)
            for layout in ("sample_major", "leaf_major"):
                leaf = torch.from_numpy(h_leaf).to(cuda)
                if layout == "leaf_major":
                    leaf = leaf.t().contiguous().t()
                root = torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=cuda)
                f(root, leaf)
                torch.cuda.synchronize()
                got = root.cpu().numpy()
                assert same(got, want), (seed, spec, layout)
