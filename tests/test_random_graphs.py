"""Randomised structure tests: small random DAGs (Sum/Prod/Power{2,3}, shared nodes, duplicate children,
factors incl. +-1 and 0-free, leaves as roots, interior roots, dead code) through
 * the allocator/scheduler replay on the CPU (every register budget), and
 * (GPU) all three back ends,
against the oracle, bit for bit."""
import os

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


@pytest.mark.gpu
def test_random_graphs_every_layout_and_mode_on_device(libfdg, cuda):
    """Every layout of the boundary (leaf-major, tile-major, row-major with contiguous and with padded rows) x evaluation and accumulation on
    random graphs of 1 ... 200 leaves -- whole tiles and a ragged last tile -- through the ISA back end: roots the oracle's bits, weighted sums
    within 1e-12 of the terms' scale.  (The linear row-major kernels' last load is half a wave wide when the leaf count is odd; graphs with
    absent roots, leaves as roots, roots shared between slots are in the mix.)"""
    import torch
    from test_tile_major import to_tiles, from_tiles
    for seed in list(range(0, 40, 3)) + list(range(1000, 1056, 5)):
        t = random_table(seed) if seed < 1000 else fuzz_table(seed)[0]
        L, R = t.n_leaf, t.n_root
        live = t.root_slot != FDG_NO_ROOT
        f = fd.compile_table(t, specialize="isa")
        for B in (64 * 3, 64 * 5 + 17):
            h_leaf = oracle.philox_uniform(B, L, seed) * 2 - 0.5
            want = oracle.eval_static(t, h_leaf, np.full((B, R), 9.0))
            w = np.random.default_rng(seed + B).uniform(0.5, 1.5, B)
            terms = np.where(np.isfinite(want), want, 0.0) * w[:, None]
            finite = np.isfinite(want).all(0) & live
            tol = 1e-12 * np.maximum(1.0, np.abs(terms).sum(0))
            dw = torch.from_numpy(w).to(cuda)
            pad = torch.full((B, L + 3), float("nan"), dtype=torch.float64, device=cuda)
            pad[:, :L] = torch.from_numpy(h_leaf).to(cuda)
            layouts = {"row-major": torch.from_numpy(h_leaf).to(cuda), "row-major padded": pad[:, :L],
                       "leaf-major": torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()}
            for lay, leaf in layouts.items():
                root = torch.full((B, R), 9.0, dtype=torch.float64, device=cuda)
                f(root, leaf)
                torch.cuda.synchronize()
                assert same(root.cpu().numpy(), want), (seed, lay, B, f.kernel_info()["last_kernel"])
                acc = torch.zeros(R, dtype=torch.float64, device=cuda)
                f.accumulate(leaf, dw, acc)
                torch.cuda.synchronize()
                a = acc.cpu().numpy()
                assert np.all(np.abs(a - terms.sum(0))[finite] <= tol[finite]), (seed, lay, B, f.kernel_info()["last_kernel"])
            tl = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
            rt = torch.full(((B + 63) // 64, R, 64), 9.0, dtype=torch.float64, device=cuda)
            f.eval_tiled(rt, tl, B)
            torch.cuda.synchronize()
            assert same(from_tiles(rt.cpu().numpy(), B, R), want), (seed, "tile-major", B)
            acc = torch.zeros(R, dtype=torch.float64, device=cuda)
            f.accumulate_tiled(tl, dw, acc, B)
            torch.cuda.synchronize()
            assert np.all(np.abs(acc.cpu().numpy() - terms.sum(0))[finite] <= tol[finite]), (seed, "tile-major", B)


def fuzz_table(seed: int):
    """Larger random DAGs for the register-pressure fuzz: up to 200 leaves and 1500 nodes, operands near or far."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 200))
    N = int(rng.choice([5, 40, 300, 1500]))
    facs = [1.0, 1.0, 1.0, -1.0, -1.0, 2.0, -0.5, 0.25, 3.0, -7.5, 1e-3, 1.0 / 3.0]
    nodes = []
    for n in range(N):
        nv = L + n
        r = rng.random()
        if r < 0.05:
            nodes.append((OP_POWER, int(rng.choice([2, 3])), [(int(rng.integers(0, nv)), float(rng.choice(facs)))]))
            continue
        op = OP_SUM if r < 0.45 else OP_PROD
        k = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 7, 30]))
        spread = float(rng.choice([3, 20, 200]))
        ch = [(int(nv - 1 - min(nv - 1, int(rng.exponential(spread)))) if rng.random() < 0.7 else int(rng.integers(0, nv)),
               float(rng.choice(facs))) for _ in range(k)]
        nodes.append((op, 0, ch))
    R = int(rng.integers(1, 8))
    roots = [int(rng.integers(0, L + N)) for _ in range(R)]
    roots[0] = L + N - 1
    return from_program(L, nodes, roots, f"fuzz_{seed}"), rng


FUZZ_SEEDS = list(range(1000, 1056))


def test_cooperative_programs_on_random_graphs(libfdg, fdgopt):
    """The cooperative variant's four programs (tests/test_next_rows.py: replay_coop) on the fuzz graphs that have a wide
    root: oracle bits including the sign of zeros, equal barrier counts, no shared slot rewritten in an epoch in which another
    wave still reads it (a node whose value is a received copy publishes that copy: its M_SEND keeps the copy's slot alive)."""
    from test_next_rows import replay_coop
    n = 0
    for seed in list(FUZZ_SEEDS) + list(range(2000, 2030)):
        t, _ = fuzz_table(seed)
        fdgopt.set("FDG_COOP_WAVES", "4" if seed % 2 else "8")
        h = capi.GraphHandle(t)
        try:
            progs, info = h.coop_program()
        except capi.FdgError as e:
            assert e.code == capi.FDG_E_UNSUPPORTED
            continue
        n += 1
        leaf = oracle.philox_uniform(24, t.n_leaf, seed) * 2 - 0.7
        want = oracle.eval_static(t, leaf)
        with np.errstate(all="ignore"):
            got = replay_coop(progs, info, leaf, t.n_root)
        live = t.root_slot != FDG_NO_ROOT
        assert same(got[:, live], want[:, live]), seed
    fdgopt.unset("FDG_COOP_WAVES")
    assert n >= 10


@pytest.mark.gpu
def test_fuzz_isa_register_budgets_on_device(libfdg, cuda, tmp_path, monkeypatch, fdgopt):
    """The optimizing back end under random register / LDS / AGPR budgets (spills through every level), value-numbering
    windows, the forget-and-recompute window, and -- every fourth seed -- the two-samples-per-lane variant: evaluation in
    both layouts bit for bit against the oracle, fused accumulation within the stated tolerance, and every listing clean
    against the emitter's wait-state table.  (tools/gpu_fuzz.py is the open-ended version of this test.)"""
    import glob
    import torch
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o700)
    for seed in FUZZ_SEEDS:
        t, rng = fuzz_table(seed)
        B = int(rng.choice([1, 63, 64, 65, 700, 140_000]))
        h_leaf = oracle.philox_uniform(B, t.n_leaf, seed) * 2 - 0.7
        want = oracle.eval_static(t, h_leaf, np.full((B, t.n_root), 9.0))
        opts = [dict(n_reg=int(rng.integers(6, 40)), n_lds=int(rng.integers(1, 30)), vn_window=int(rng.choice([1, 20, 200, 1000]))),
                dict(n_reg=int(rng.integers(30, 120)), n_lds=int(rng.integers(1, 80)), n_acc=int(rng.integers(1, 124)))]
        if seed % 4 == 0:
            fdgopt.set("FDG_ISA_W2", "1")
            opts.append(None)
        if seed % 3 == 0:
            fdgopt.set("FDG_REMAT_WINDOW", str(int(rng.choice([8, 60, 400]))))
        if seed % 2 == 1:                  # the cooperative variant wherever the graph has a wide root sum (it then takes the leaf-major calls)
            fdgopt.set("FDG_ISA_COOP", "1")
            fdgopt.set("FDG_COOP_WAVES", "4" if seed % 4 == 1 else "8")
        for opt in opts:
            f = fd.compile_table(t, specialize="isa", opt=opt, cache_dir=str(cache), flags=capi.FDG_SPEC_KEEP_SOURCE)
            for layout in ("leaf_major", "sample_major"):
                leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t() if layout == "leaf_major" else torch.from_numpy(h_leaf).to(cuda)
                # (even seeds: the roots as a Julia column-major matrix too -- with B a multiple of 16 the streaming kernels run)
                root = (torch.full((t.n_root, B), 9.0, dtype=torch.float64, device=cuda).t() if layout == "leaf_major" and seed % 2 == 0
                        else torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=cuda))
                f(root, leaf)
                torch.cuda.synchronize()
                assert same(root.cpu().numpy(), want), (seed, opt, layout, B)
            w = torch.rand(B, dtype=torch.float64, device=cuda)
            acc = f.accumulate(leaf, w)
            torch.cuda.synchronize()
            if np.isfinite(want).all():
                live = t.root_slot != FDG_NO_ROOT
                wr = want * w.cpu().numpy()[:, None]
                assert np.all(np.abs(acc.cpu().numpy() - wr.sum(0))[live] <= 1e-12 * np.maximum(1.0, np.abs(wr).sum(0))[live]), (seed, opt)
        fdgopt.unset("FDG_ISA_W2")
        fdgopt.unset("FDG_REMAT_WINDOW")
        fdgopt.unset("FDG_ISA_COOP")
        fdgopt.unset("FDG_COOP_WAVES")
        for lst in glob.glob(str(cache / "*.s")):
            n, rep = capi.isa_check_hazards(open(lst).read())
            assert n == 0, (seed, rep)
            os.remove(lst)


# --------------------------------------------------------------------------- #
# one-kernel Monte-Carlo step on random graphs over random leaf tables
# --------------------------------------------------------------------------- #
def random_leaf_tables(seed: int, L: int):
    """A random partition in the shape FrontEnds.leafstates produces (frontends.jl:178-232): fermionic leaves with
    green_derive orders 0..5 and interaction leaves with counter-term orders 0..6 over a random loop basis."""
    rng = np.random.default_rng(1000 + seed)
    n_loop, n_tau, n_basis = int(rng.integers(1, 5)), int(rng.integers(1, 6)), int(rng.integers(1, 9))
    basis = rng.choice([-1.0, 0.0, 0.0, 1.0, 1.0, 0.5], size=(n_basis, n_loop))
    for r in range(n_basis):
        if not basis[r].any():
            basis[r, int(rng.integers(0, n_loop))] = 1.0
    ty = rng.choice([1, 1, 1, 2, 2, 0], size=L).astype(np.int32)     # 0: a leaf without a formula (value 1.0)
    order = np.where(ty == 1, rng.integers(0, 6, size=L), rng.integers(0, 7, size=L)).astype(np.int32)
    order[rng.random(L) < 0.5] = 0
    return dict(leaf_type=ty, leaf_order=order, tau_in=rng.integers(1, n_tau + 1, size=L).astype(np.int32),
                tau_out=rng.integers(1, n_tau + 1, size=L).astype(np.int32), loop_index=rng.integers(1, n_basis + 1, size=L).astype(np.int32),
                basis=basis, n_tau=n_tau, n_loop=n_loop)


def leaves_table(L):
    from feynmandiagram_jl_amd.nodetable import NodeTable
    return NodeTable(L, np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0),
                     np.arange(L, dtype=np.uint32), "leaves")


def check_leaves(z, got, K, T, kF, beta, lam):
    """leaf values against the oracle: 1e-12 of the largest Leibniz term for derivatives, 1e-13 relative otherwise"""
    args = (z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"])
    want = oracle.leaf_values(*args, K, T, kF, beta, lam)
    q2 = (np.einsum("bjd,nj->bnd", K, z["basis"]) ** 2).sum(axis=2)
    for i in range(len(z["leaf_type"])):
        if z["leaf_type"][i] == 1 and z["leaf_order"][i] > 0:
            tau = T[:, z["tau_out"][i] - 1] - T[:, z["tau_in"][i] - 1]
            scale = oracle.green_derive_scale(tau, q2[:, z["loop_index"][i] - 1] - kF * kF, beta, int(z["leaf_order"][i]))
            assert np.all(np.abs(got[:, i] - want[:, i]) <= 1e-12 * scale), (i, int(z["leaf_order"][i]))
        elif z["leaf_type"][i] == 0:
            assert np.all(got[:, i] == 1.0), i
        else:
            assert np.all(np.abs(got[:, i] - want[:, i]) <= 1e-13 * np.abs(want[:, i])), (i, int(z["leaf_type"][i]), int(z["leaf_order"][i]))


MC_SEEDS = list(range(int(os.environ.get("FDG_TEST_MC_SEEDS", "16"))))      # more seeds for a soak: FDG_TEST_MC_SEEDS=400


@pytest.mark.parametrize("seed", MC_SEEDS)
def test_random_mc_program_replays(libfdg, seed):
    """fdg_graph_mc_program on random graphs over random leaf tables, replayed in numpy: the graph part is exact (the
    roots are the oracle's graph applied to the leaves of the same formulas, read out through the leaves-as-roots
    program), the leaves agree with the oracle's within the leaf kernels' tolerance."""
    from test_next_rows import replay_mc
    t = random_table(seed)
    z = random_leaf_tables(seed, t.n_leaf)
    kF, beta, lam = 1.3, float(np.random.default_rng(seed).choice([0.7, 3.0, 25.0])), 0.9
    dim = 3
    tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, z["n_tau"], kF, beta, lam)
    rng = np.random.default_rng(seed)
    B = 24
    K = rng.uniform(-2, 2, (B, z["n_loop"], dim)); T = rng.uniform(0, beta, (B, z["n_tau"]))
    X = np.concatenate([K.reshape(B, -1), T], axis=1)
    budget = dict(n_reg=int(rng.integers(6, 14)), n_lds=int(rng.integers(1, 4)), n_acc=int(rng.integers(0, 3)), vn_window=int(rng.choice([0, 5, 200])))
    ops, nr, nl, nm = capi.GraphHandle(leaves_table(t.n_leaf)).mc_program(tab, **budget)
    leaves = replay_mc(ops, nr, nl, nm, budget["n_acc"], X, t.n_leaf)
    check_leaves(z, leaves, K, T, kF, beta, lam)
    ops, nr, nl, nm = capi.GraphHandle(t).mc_program(tab, **budget)
    got = replay_mc(ops, nr, nl, nm, budget["n_acc"], X, t.n_root)
    want = oracle.eval_static(t, leaves)
    live = t.root_slot != FDG_NO_ROOT
    assert same(got[:, live], want[:, live]), (seed, budget)
    # kF, beta, lambda reach the kernel as arguments: the ops that use them are tagged (param 1..4 = -kF^2, beta, -beta,
    # lambda) -- substituting other values there gives the program of the other parameter set
    kF2, beta2, lam2 = 0.8, 1.7 * beta, 0.4
    T2 = T * (beta2 / beta)
    ops, nr, nl, nm = capi.GraphHandle(leaves_table(t.n_leaf)).mc_program(tab, **budget)
    assert set(np.unique(ops["param"])) <= {0, 1, 2, 3, 4}
    ops2 = ops.copy()
    for tag, val in ((1, -(kF2 * kF2)), (2, beta2), (3, -beta2), (4, lam2)):
        ops2["imm"][ops["param"] == tag] = val
    leaves2 = replay_mc(ops2, nr, nl, nm, budget["n_acc"], np.concatenate([K.reshape(B, -1), T2], axis=1), t.n_leaf)
    check_leaves(z, leaves2, K, T2, kF2, beta2, lam2)


@pytest.mark.gpu
def test_random_mc_step_on_device(libfdg, cuda, monkeypatch, fdgopt):
    """The same statements for the kernels: eval and accumulate of the one-kernel route on random graphs and tables."""
    import torch
    fdgopt.set("FDG_MC_ROUTE", "isa")
    st = torch.cuda.current_stream().cuda_stream
    for seed in MC_SEEDS:
        t = random_table(seed)
        L, R = t.n_leaf, t.n_root
        z = random_leaf_tables(seed, L)
        rng = np.random.default_rng(seed)
        kF, beta, lam = 1.3, float(rng.choice([0.7, 3.0, 25.0])), 0.9
        dim, n_k, n_tau = 3, z["n_loop"] * 3, z["n_tau"]
        tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
        B = int(rng.choice([1, 65, 1000]))
        K = rng.uniform(-2, 2, (B, z["n_loop"], dim)); T = rng.uniform(0, beta, (B, n_tau))
        dK = torch.from_numpy(np.ascontiguousarray(K.reshape(B, n_k).T)).to(cuda)
        dT = torch.from_numpy(np.ascontiguousarray(T.T)).to(cuda)
        gl = fd.compile_table(leaves_table(L), specialize="isa"); gl.handle.specialize_fused(tab)
        d_leaves = torch.zeros((B, L), dtype=torch.float64, device=cuda)
        gl.handle.mc_eval_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, d_leaves.data_ptr(), L, 1, B, st)
        torch.cuda.synchronize()
        leaves = d_leaves.cpu().numpy()
        check_leaves(z, leaves, K, T, kF, beta, lam)
        g = fd.compile_table(t, specialize="isa"); g.handle.specialize_fused(tab)
        root = torch.full((B, R), 9.0, dtype=torch.float64, device=cuda)
        g.handle.mc_eval_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, root.data_ptr(), R, 1, B, st)
        torch.cuda.synchronize()
        want = oracle.eval_static(t, leaves, np.full((B, R), 9.0))
        got = root.cpu().numpy()
        assert same(got, want), seed
        live = t.root_slot != FDG_NO_ROOT
        w = torch.rand(B, dtype=torch.float64, device=cuda)
        acc = torch.zeros(R, dtype=torch.float64, device=cuda)
        g.handle.mc_accumulate_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, w.data_ptr(), acc.data_ptr(), B, st)
        torch.cuda.synchronize()
        wr = np.where(live[None, :], got, 0.0) * w.cpu().numpy()[:, None]
        fin = np.isfinite(wr).all(axis=0) & live
        assert np.all(np.abs(acc.cpu().numpy() - wr.sum(0))[fin] <= 1e-12 * np.maximum(1.0, np.abs(wr).sum(0))[fin]), seed


def power_table(seed: int, N: int = 300):
    """A random graph whose Power nodes take every kind of exponent: literal_pow (2, 3, -1, -2) and pow_body (4, 5, 7, -3, -4)."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(20, 120))
    facs = [1.0, 1.0, -1.0, 2.0, -0.5, 0.25, 3.0, 1.0 / 3.0]
    nodes = []
    for n in range(N):
        nv = L + n
        r = rng.random()
        if r < 0.10:
            nodes.append((OP_POWER, int(rng.choice([2, 3, 4, 5, 7, -1, -2, -3, -4])), [(int(rng.integers(0, nv)), float(rng.choice(facs)))]))
            continue
        op = OP_SUM if r < 0.55 else OP_PROD
        k = int(rng.choice([2, 2, 3, 4, 9]))
        ch = [(int(nv - 1 - min(nv - 1, int(rng.exponential(25)))) if rng.random() < 0.7 else int(rng.integers(0, nv)), float(rng.choice(facs))) for _ in range(k)]
        nodes.append((op, 0, ch))
    R = int(rng.choice([2, 7, 30]))
    roots = [int(rng.integers(L, L + N)) for _ in range(R)]
    roots[0] = L + N - 1
    return from_program(L, nodes, roots, f"power_{seed}")


@pytest.mark.parametrize("seed", range(6))
def test_pow_body_powers_in_every_kernel_variant_assemble(libfdg, tmp_path, seed):
    """Round 5 (found by the device fuzz once it drew exponents outside {2, 3}): pow_body's temporaries (four register pairs for the
    correctly rounded division) and the cooperative section's own registers together ran past v255 -- the assembler refused the code
    object.  Every variant the handle builds (one-wave, accumulate, streaming, row-major, linear row-major, cooperative forced on) must
    assemble, keep the hazard table, and the allocated one-wave program must replay to the oracle's bits."""
    os.chmod(tmp_path, 0o700)
    t = power_table(seed)
    f = fd.compile_table(t, specialize="isa", cache_dir=str(tmp_path), flags=capi.FDG_SPEC_KEEP_SOURCE,
                         options={"FDG_ISA_COOP": "1", "FDG_CACHE_RO_DIR": ""})
    ki = f.kernel_info()
    asm = [x for x in os.listdir(tmp_path) if x.endswith(".s")]
    assert asm
    text = open(os.path.join(tmp_path, asm[0])).read()
    assert capi.isa_check_hazards(text)[0] == 0
    assert "v_div_fixup_f64" in text                                   # a negative exponent: the correctly rounded reciprocal
    leaf = oracle.philox_uniform(9, t.n_leaf, seed) + 0.4
    ops, nr, nl, nm = f.handle.opt_program()
    with np.errstate(all="ignore"):
        got = replay(ops, nr, nl, nm, f.handle.last_n_acc, leaf, t.n_root)
    assert same(got, oracle.eval_static(t, leaf))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("coop", [False, True])
def test_pow_body_powers_on_device_in_every_layout(libfdg, cuda, seed, coop):
    """The same graphs on the device: leaf-major, row-major and tile-major evaluation bit for bit (NaNs in place, zeros signed) and fused
    accumulation within 1e-12 of the terms' scale, with the cooperative variant forced on and left to the library."""
    import torch
    from test_tile_major import to_tiles, from_tiles
    t = power_table(seed)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa", options={"FDG_ISA_COOP": "1"} if coop else None)
    for B in (64 * 5 + 17, 4099):
        h_leaf = oracle.philox_uniform(B, L, seed) + 0.4
        with np.errstate(all="ignore"):
            want = oracle.eval_static(t, h_leaf)
        lm = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
        assert same(f(None, lm).cpu().numpy(), want), ("leaf-major", f.kernel_info()["last_kernel"])
        rm = torch.from_numpy(h_leaf).to(cuda)
        assert same(f(None, rm).cpu().numpy(), want), ("row-major", f.kernel_info()["last_kernel"])
        tm = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
        rt = f.eval_tiled(None, tm, B)
        assert same(from_tiles(rt.cpu().numpy(), B, R), want), ("tile-major", f.kernel_info()["last_kernel"])
        if np.isfinite(want).all():
            w = np.random.default_rng(seed).uniform(0.5, 1.5, B)
            acc = f.accumulate_tiled(tm, torch.from_numpy(w).to(cuda), None, B).cpu().numpy()
            terms = want * w[:, None]
            assert np.all(np.abs(acc - terms.sum(0)) <= 1e-12 * np.maximum(1.0, np.abs(terms).sum(0)))
