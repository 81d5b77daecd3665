"""The restated Parquet front end (feynmandiagram.jl_amd/producers/parquet.py), pinned by the reference's own tests of it
(test/front_end.jl:120-700: parameters, partitions, index helpers, filters, the diagram counts of the self-energy, the
validity of Green's functions), by the optimized 2-loop graph the reference renders in assets/sigma_o2.svg (SURVEY.md
Appendix A) and by value invariance under ``optimize!``.  No GPU."""
import numpy as np
import pytest

import oracle
from feynmandiagram_jl_amd import fixtures, workloads
from feynmandiagram_jl_amd.producers import optimize, parquet as pq
from feynmandiagram_jl_amd.producers.gv import mirror_symmetrize
from feynmandiagram_jl_amd.graph import Graph
from feynmandiagram_jl_amd.lowering import lower
from feynmandiagram_jl_amd.producers.parquet import (ChargeCharge, DiagPara, Dynamic, Girreducible, GreenDiag, Instant, Interaction, NoFock,
                                           NoHartree, SigmaDiag, Ver4Diag, reconstruct)


def all_ones(graphs):
    t, _, _ = lower(list(graphs))
    return oracle.eval_static(t, np.ones((1, t.n_leaf)))[0]


# ---- test/front_end.jl:126-146 ------------------------------------------------------------------------------------
def test_parameter_equality():
    p = DiagPara(type=Ver4Diag, innerLoopNum=1)
    q = DiagPara(type=Ver4Diag, innerLoopNum=2)
    a = DiagPara(type=Ver4Diag, innerLoopNum=2)
    assert p != q and q == a
    assert a != reconstruct(a, transferLoop=(0.0, 0.0, 0.0))
    assert a != reconstruct(a, interaction=())
    # a reconstructed parameter keeps firstLoopIdx, totalLoopNum ... of the old type
    assert reconstruct(a, type=SigmaDiag) != DiagPara(type=SigmaDiag, innerLoopNum=2)
    # defaults of the keyword constructor (parquet.jl:104-125)
    s = DiagPara(type=SigmaDiag, innerLoopNum=4)
    assert (s.firstLoopIdx, s.totalLoopNum, s.firstTauIdx, s.totalTauNum) == (2, 5, 1, 4)
    v = DiagPara(type=Ver4Diag, innerLoopNum=2, interaction=(Interaction(ChargeCharge, (Instant, Dynamic)),))
    assert (v.firstLoopIdx, v.totalLoopNum, v.totalTauNum) == (4, 5, 6)


# ---- test/front_end.jl:149-156 (compared as sets there too) ---------------------------------------------------------
def test_ordered_partition():
    as_set = lambda p: {tuple(x) for x in p}
    assert as_set(pq.orderedPartition(5, 2)) == {(4, 1), (1, 4), (2, 3), (3, 2)}
    assert as_set(pq.orderedPartition(3, 2, 0)) == {(3, 0), (0, 3), (1, 2), (2, 1)}
    p = pq.orderedPartition(2, 4, 0)
    assert len(p) == len(as_set(p)) == 10 and all(sum(x) == 2 for x in p)
    assert pq.orderedPartition(0, 2, 0) == [[0, 0]]


# ---- test/front_end.jl:158-183 ----------------------------------------------------------------------------------------
def test_find_first_indices():
    for partition, first, want in (([1, 1, 2, 1], 1, [1, 2, 3, 5]), ([1, 1, 2, 1], 0, [0, 1, 2, 4]), ([1, 0, 2, 0], 1, [1, 2, 2, 4]), ([1], 1, [1])):
        idx, total = pq.findFirstLoopIdx(partition, first)
        assert idx == want and total == sum(partition) + first - 1
    kinds = [Ver4Diag, GreenDiag, Ver4Diag, GreenDiag]
    assert pq.findFirstTauIdx([1, 1, 2, 1], kinds, 1, 1)[0] == [1, 3, 4, 7]
    assert pq.findFirstTauIdx([1, 1, 2, 1], kinds, 0, 1)[0] == [0, 2, 3, 6]
    assert pq.findFirstTauIdx([1, 0, 2, 0], kinds, 1, 1)[0] == [1, 3, 3, 6]


# ---- test/front_end.jl:185-219 ----------------------------------------------------------------------------------------
def test_filters():
    assert pq.isValidG([Girreducible], 0) and not pq.isValidG([Girreducible], 1) and not pq.isValidG([Girreducible], 2)
    assert pq.isValidG([NoFock], 0) and pq.isValidG([NoFock], 1)
    assert not pq.isValidG([NoFock, NoHartree], 1) and pq.isValidG([NoFock, NoHartree], 2)
    for n, sub, want in ((0, True, False), (1, True, False), (2, True, False), (0, False, False), (1, False, True), (2, False, True)):
        assert pq.isValidSigma([Girreducible], n, sub) == want
    assert not pq.isValidSigma([NoFock], 0, True)
    assert pq.isValidSigma([NoFock], 1, True)
    assert not pq.isValidSigma([NoFock, NoHartree], 1, True)
    assert pq.isValidSigma([NoFock, NoHartree], 2, True)
    assert not pq.isValidSigma([NoFock], 0, False)
    assert pq.isValidSigma([NoFock], 1, False) and pq.isValidSigma([NoFock, NoHartree], 1, False) and pq.isValidSigma([NoFock], 2, False)


# ---- test/front_end.jl:600-652: the number of self-energy diagrams of the G^2 v expansion ------------------------------
@pytest.mark.parametrize("loops", [1, 2, 3, 4])
def test_sigma_diagram_counts(loops):
    para = DiagPara(type=SigmaDiag, hasTau=True, innerLoopNum=loops, totalLoopNum=loops + 1, totalTauNum=loops, isFermi=False, spin=2,
                    firstLoopIdx=2, firstTauIdx=1, filter=(NoHartree, Girreducible), interaction=(Interaction(ChargeCharge, Instant),),
                    extra=pq.ParquetBlocks(phi=(pq.PHEr, pq.PPr), ppi=(pq.PHr, pq.PHEr)))
    extK = [1.0] + [0.0] * loops
    rows = pq.sigma(para, extK, False)
    merged = pq.mergeby([r["diagram"] for r in rows])           # `mergeby(diag)` of the test: one Sum over the rows
    assert len(merged) == 1
    num = all_ones(merged)[0]
    want = {1: 1, 2: 3, 3: 18, 4: 171}[loops]                   # 1, 1 + spin, 4 + 5 spin + spin^2, 27 + 40 spin + 14 spin^2 + spin^3
    assert pq.count_sigma_G2v(loops, 2) == want
    assert num * (-1) ** loops == want
    assert all(r["extT"][0] == para.firstTauIdx for r in rows)


@pytest.mark.parametrize("loops", [1, 2, 3, 4])
def test_sigma_diagram_counts_with_dynamic_interactions(loops):
    """The same count with a Dynamic interaction (every line carries two time indices instead of one: same diagrams), and with
    an Instant + Dynamic one (every one of the n interaction lines is either: 2^n times as many)."""
    for types, mult in (((Dynamic,), 1), ((Instant, Dynamic), 2 ** loops)):
        para = DiagPara(type=SigmaDiag, innerLoopNum=loops, isFermi=False, filter=(NoHartree, Girreducible), interaction=(Interaction(ChargeCharge, types),))
        rows = pq.sigma(para, [1.0] + [0.0] * loops, False)
        assert sum(all_ones([r["diagram"] for r in rows])) * (-1) ** loops == mult * pq.count_sigma_G2v(loops, 2)
        assert all(r["extT"][0] == para.firstTauIdx for r in rows)


@pytest.mark.parametrize("loops", [2, 3, 4, 5, 6])
def test_fermionic_sums_agree_with_the_gv_catalogs(loops):
    """Two front ends of the reference for the same quantity: the Parquet builder (fermionic signs, spin 2, NoHartree) and the
    GV catalogs groups_sigma/Sigma<n>_0_0.diag, whose sums of SymFactor * SpinFactor per external-time group are computed
    from the file text (tests/golden/make_gv_tables.py: [dynamic, instant] = 2: 1 / -1, 3: -5 / 1, 4: 21 / 3, 5: -77 / -31,
    6: 233 / 167).  With all leaves 1 the instantaneous part and the sum of the dynamic parts of the Parquet self-energy
    equal the catalog's sums times -1 (the two front ends differ by one overall sign) -- at 5 and 6 loops this includes the
    fully irreducible vertices read from the vertex catalogs and moved into the caller's basis."""
    catalog = {2: (1.0, -1.0), 3: (-5.0, 1.0), 4: (21.0, 3.0), 5: (-77.0, -31.0), 6: (233.0, 167.0)}[loops]
    rows = pq.build(DiagPara(type=SigmaDiag, innerLoopNum=loops, filter=(NoHartree,)))
    assert rows[0]["type"] == Instant and all(r["type"] == Dynamic for r in rows[1:])
    v = all_ones([r["diagram"] for r in rows])
    assert (float(v[1:].sum()), float(v[0])) == (-catalog[0], -catalog[1])
    if loops >= 4:                                   # the shipped GV tables say the same
        g = oracle.eval_static(workloads.get(f"gv_sigma{loops}"), np.ones((1, workloads.get(f"gv_sigma{loops}").n_leaf)))[0]
        assert sorted(g) == sorted(catalog)


# ---- test/front_end.jl:701-755: the 3-point vertex; :758-826: the polarization (three variants) -------------------------
@pytest.mark.parametrize("loops", [1, 2, 3])
def test_vertex3_diagram_counts(loops):
    para = DiagPara(type=pq.Ver3Diag, innerLoopNum=loops, isFermi=False, hasTau=True, filter=(NoHartree, Girreducible, pq.Proper),
                    interaction=(Interaction(ChargeCharge, Instant),))
    Q, KinL = [0.0] * para.totalLoopNum, [0.0] * para.totalLoopNum
    Q[0], KinL[1] = 1.0, 1.0
    rows = pq.vertex3(para, [Q, KinL])
    num = all_ones([pq.mergeby(rows)[0]["diagram"]])[0]
    assert num * (-1) ** loops == pq.count_ver3_G2v(loops, 2) == {1: 1, 2: 10, 3: 109}[loops]


@pytest.mark.parametrize("loops", [1, 2, 3, 4])
def test_polarization_diagram_counts(loops):
    def polar(filter):
        para = DiagPara(type=pq.PolarDiag, innerLoopNum=loops, isFermi=False, hasTau=True, filter=filter, interaction=(Interaction(ChargeCharge, Instant),))
        Q = [1.0] + [0.0] * (para.totalLoopNum - 1)
        return para, pq.polarization(para, Q)

    sign = 2 * (-1) ** (loops - 1)                       # num * spin * (-1)^(n-1)
    para, rows = polar((NoHartree, Girreducible))        # G^2 v expansion
    assert all_ones([pq.mergeby(rows)[0]["diagram"]])[0] * sign == pq.count_polar_G2v(loops, 2) == {1: 2, 2: 2, 3: 20, 4: 218}[loops]
    para, rows = polar((NoHartree, NoFock))              # g^2 v expansion: Green's functions carry self-energy insertions
    assert all_ones([pq.mergeby(rows)[0]["diagram"]])[0] * sign == {1: 2, 2: 2, 3: 32, 4: 326}[loops]
    assert rows[0]["response"] == pq.UpUp
    assert all_ones([rows[0]["diagram"]])[0] * sign == {1: 2, 2: 2, 3: 28, 4: 274}[loops]        # <n_up n_up> alone
    assert all(r["extT"] == (para.firstTauIdx, para.firstTauIdx + 1) for r in rows)
    # a parameter that names Proper explicitly builds as well (test/front_end.jl:785)
    if loops == 1:
        assert len(polar((pq.Proper, NoHartree, NoFock))[1]) == 1


# ---- test/front_end.jl:654-699 ----------------------------------------------------------------------------------------
def test_green_validity():
    def buildG(loops, extT, filter):
        para = DiagPara(type=GreenDiag, hasTau=True, innerLoopNum=loops, isFermi=True, spin=2, filter=filter,
                        interaction=(Interaction(ChargeCharge, Instant),))
        extK = [1.0] + [0.0] * (para.totalLoopNum - 1)
        return pq.green(para, extK, extT) if pq.isValidG(para) else None

    assert isinstance(buildG(0, (1, 2), (NoHartree, Girreducible)), Graph)
    assert buildG(1, (1, 2), (NoHartree, Girreducible)) is None
    assert buildG(2, (1, 2), (NoHartree, Girreducible)) is None
    assert isinstance(buildG(0, (1, 2), (NoHartree, NoFock)), Graph)
    assert buildG(1, (1, 2), (NoHartree, NoFock)) is None
    assert isinstance(buildG(2, (1, 2), (NoHartree, NoFock)), Graph)


# ---- README.md:55-72 + assets/sigma_o2.svg ------------------------------------------------------------------------------
def _canonical(t):
    """The node table up to the order of a node's operands: (op, power, sorted (child signature, factor)) per value."""
    sig = {}
    for i in range(t.n_leaf):
        sig[i] = (-1, i, ())
    for n in range(t.n_node):
        lo, hi = int(t.child_off[n]), int(t.child_off[n + 1])
        ch = sorted((repr(sig[int(t.child_idx[e])]), float(t.child_fac[e])) for e in range(lo, hi))
        sig[t.n_leaf + n] = (int(t.op[n]), int(t.power[n]), tuple(ch))
    return [sig[int(r)] for r in t.root_slot]


def test_two_loop_self_energy_is_the_graph_of_the_reference_rendering():
    """``Parquet.build(DiagPara(type=SigmaDiag, innerLoopNum=2, hasTau=true, filter=[NoHartree]))`` + ``optimize!``.
    The rendering shows the graph after chains left behind by the merge of linear combinations were flattened as well
    (a second ``optimize!``: 18 nodes; one pass leaves one unary node, 19); graphviz does not keep the order of a
    node's operands, so the comparison is up to that order -- the leaf numbering, which the rendering prints
    (G1..G8), is compared exactly."""
    para = DiagPara(type=SigmaDiag, innerLoopNum=2, hasTau=True, filter=(NoHartree,))
    rows = pq.build(para)
    assert [(r["type"], r["extT"]) for r in rows] == [(Instant, (1, 1)), (Dynamic, (1, 2))]        # README.md:66-68
    graphs = [r["diagram"] for r in rows]
    before = all_ones(graphs)
    optimize.optimize_(graphs)
    once, _, _ = lower(graphs)
    assert (once.n_leaf, once.n_node, once.n_root) == (8, 19, 2)
    optimize.optimize_(graphs)
    mine, leafmap, _ = lower(graphs)
    ref, ref_leafmap, _ = lower(fixtures.sigma2_graphs()[0])
    assert (mine.n_leaf, mine.n_node, mine.n_root) == (ref.n_leaf, ref.n_node, ref.n_root) == (8, 18, 2)
    assert np.array_equal(mine.op, ref.op) and np.array_equal(mine.child_off, ref.child_off) and np.array_equal(mine.root_slot, ref.root_slot)
    assert _canonical(mine) == _canonical(ref)
    kinds = ["G" if type(leafmap[i + 1].properties).__name__ == "BareGreenId" else "V" for i in range(8)]
    assert kinds == [ref_leafmap[i + 1].name for i in range(8)] == ["G", "G", "V", "G", "V", "V", "G", "G"]
    assert list(before) == list(all_ones(graphs)) == [1.0, -1.0]
    # the same workload as a named table
    t = workloads.get("parquet_sigma2")
    assert np.array_equal(t.child_idx, once.normalized().child_idx)


@pytest.mark.parametrize("name, sizes", [("parquet_sigma3", (27, 156, 3)), ("parquet_sigma4", (84, 1325, 4)), ("parquet_sigma4_dyn", (175, 4819, 7)),
                                         ("parquet_sigma4_insdyn", (312, 20147, 8)), ("parquet_sigma4_taylor2", (116, 7421, 12))])
def test_four_loop_self_energy_tables(name, sizes):
    t = workloads.get(name)
    assert (t.n_leaf, t.n_node, t.n_root) == sizes
    rng = np.random.default_rng(5)
    x = rng.uniform(0.5, 1.5, size=(3, t.n_leaf))
    assert np.array_equal(oracle.eval_static(t, x), oracle.eval_interp(t, x)) or np.allclose(oracle.eval_static(t, x), oracle.eval_interp(t, x), rtol=1e-12)


def test_optimize_keeps_the_value_of_the_four_loop_self_energy():
    for types in ((Instant,), (Dynamic,), (Instant, Dynamic)):
        para = DiagPara(type=SigmaDiag, innerLoopNum=4, hasTau=True, filter=(NoHartree,), interaction=(Interaction(ChargeCharge, types),))
        rows = pq.build(para)
        assert all(r["extT"][0] == 1 for r in rows)
        graphs = [r["diagram"] for r in rows]
        raw, lm_raw, _ = lower(graphs)
        # leaves with equal identities get equal values, as they do after the merge of duplicated leaves
        rng = np.random.default_rng(9)
        val = {}
        x = np.array([[val.setdefault(lm_raw[i + 1].properties.equiv_key(), rng.uniform(0.5, 1.5)) for i in range(raw.n_leaf)]])
        want = oracle.eval_static(raw, x)[0]
        optimize.optimize_(graphs)
        opt, lm, _ = lower(graphs)
        assert opt.n_leaf == len(val)
        y = np.array([[val[lm[i + 1].properties.equiv_key()] for i in range(opt.n_leaf)]])
        got = oracle.eval_static(opt, y)[0]
        assert np.allclose(got, want, rtol=1e-11, atol=0)


def test_taylor_expansion_of_the_parquet_self_energy_satisfies_the_series_identity():
    """Config 4 on the real graph: f(V0 + x V1 + x^2 V2) = c0 + x c1 + x^2 c2 + O(x^3), checked independently of the
    Taylor restatement (as tests/golden/make_gv_tables.py does for the GV graphs)."""
    from feynmandiagram_jl_amd.producers import gv, taylor
    graphs, rows = workloads.parquet_graphs("parquet_sigma3")
    t0, lm0, _ = lower(graphs)
    d = taylor.taylorAD(graphs, [2], [lambda pr: isinstance(pr, gv.BareInteractionId)])
    allg = [g for o in sorted(d) for g in d[o]]
    optimize.optimize_(allg)
    t, lm, _ = lower(allg)
    R = len(rows)
    assert t.n_root == 3 * R
    key0 = {lm0[i + 1].properties.equiv_key(): i for i in range(t0.n_leaf)}
    base = [key0[lm[i + 1].properties.equiv_key()] for i in range(t.n_leaf)]
    dord = [int(lm[i + 1].orders[0]) if len(lm[i + 1].orders) == 1 else 0 for i in range(t.n_leaf)]
    rng = np.random.default_rng(3)
    v = rng.uniform(0.5, 1.5, size=(3, t0.n_leaf))
    is_v = np.array([isinstance(lm0[i + 1].properties, gv.BareInteractionId) for i in range(t0.n_leaf)])
    x = 1e-3
    f_x = oracle.eval_static(t0, (v[0] + np.where(is_v, x * v[1] + x * x * v[2], 0.0))[None, :])[0]
    c = oracle.eval_static(t, np.array([[v[dord[i], base[i]] for i in range(t.n_leaf)]]))[0].reshape(3, R)
    series = c[0] + x * c[1] + x * x * c[2]
    scale = np.abs(c).sum(axis=0) + 1.0
    assert np.all(np.abs(series - f_x) <= 1e-7 * scale)
    assert np.all(np.abs(c[0] + x * c[1] - f_x) >= np.abs(series - f_x))


# ---- the fully irreducible vertex from the GV catalogs (parquet.jl:216-231, vertex4.jl:112-120, operation.jl:178-257) -------
REF_GV = "/root/reference/src/frontend/GV_diagrams"


def _catalog_sums(c):
    n = len(c["symfactor"])
    di = sum(float(c["symfactor"][d]) * float(c["spin"][d][c["diex"][d] == 0].sum()) for d in range(n))
    ex = sum(float(c["symfactor"][d]) * float(c["spin"][d][c["diex"][d] == 1].sum()) for d in range(n))
    return [di + ex, di]


@pytest.mark.parametrize("order, want", [(3, [0.0, 2.0]), (4, [0.0, -26.0])])
def test_irreducible_vertex_catalogs(order, want):
    """(UpUp, UpDown) of ``diagsGV_ver4(order, channels=[Alli])``: all leaves 1 => the catalog's own sums of
    SymFactor * SpinFactor (direct terms for UpDown, direct + exchange for UpUp), computed from the numbers directly."""
    import os
    from feynmandiagram_jl_amd.producers import gv
    graphs = pq.get_ver4I(order)
    assert [g.properties.response for g in graphs] == [pq.UpUp, pq.UpDown] and all(g.properties.channel == pq.Alli for g in graphs)
    assert list(all_ones(graphs)) == want
    c = dict(np.load(os.path.join(workloads.DATA, f"vertex4I{order}.npz")))
    assert _catalog_sums(c) == want
    if os.path.isdir(REF_GV):                   # the shipped arrays are the catalog's numbers
        ref = gv.parse_vertex4_catalog(f"{REF_GV}/groups_vertex4/Vertex4I{order}_0_0.diag")
        assert all(np.array_equal(ref[k], c[k]) for k in ref)


def test_irreducible_vertex_enters_the_parquet_equations_with_the_right_multiplicity(monkeypatch):
    """The 5-loop polarization is the first object whose sub-vertices reach 3 loops.  The reference's count functions go
    that far (count_polar_g2v_noFock_upup / _updown(5) = 3586 / 844, benchmark/diagram_count.jl:82-118) although its test
    stops at 4 loops: the catalog's factors carry fermionic signs, so counting needs their absolute values -- with those
    the numbers are met exactly; with the signs as they are, the catalog vertex contributes (0, 4) instead of (168, 84)."""
    import functools
    import os
    from feynmandiagram_jl_amd.producers import gv
    para = DiagPara(type=pq.PolarDiag, innerLoopNum=5, isFermi=False, hasTau=True, filter=(NoHartree, NoFock), interaction=(Interaction(ChargeCharge, Instant),))
    Q = [1.0] + [0.0] * (para.totalLoopNum - 1)
    with_signs = all_ones([r["diagram"] for r in pq.polarization(para, Q)]) * 2
    assert list(with_signs) == [3418.0, 764.0]

    def bosonic(order):
        c = dict(np.load(os.path.join(workloads.DATA, f"vertex4I{order}.npz")))
        c["spin"], c["symfactor"] = np.abs(c["spin"]), np.abs(c["symfactor"])
        return gv.read_vertex4diagrams(c, 0.0, (NoHartree,), (pq.Alli,))

    monkeypatch.setattr(pq, "get_ver4I", functools.lru_cache(maxsize=None)(bosonic))
    rows = pq.polarization(para, Q)
    assert [r["response"] for r in rows] == [pq.UpUp, pq.UpDown]
    assert list(all_ones([r["diagram"] for r in rows]) * 2) == [3586.0, 844.0]


def test_catalog_vertex_in_the_callers_basis():
    """``update_extKT``: every propagator's momentum in the caller's basis is one linear image of its momentum in the
    catalog's basis (up to the mirror symmetry), the three external loops go to the caller's legs, the inner loops to
    positions of their own; time indices shift by firstTauIdx - 1; the copy shares no node with the cached graphs."""
    # a 3-loop sub-vertex of a bubble whose own loop is number 4: inner loops 5..7 of 7, legs built from loops 1, 2 and 4
    para = DiagPara(type=Ver4Diag, innerLoopNum=3, firstLoopIdx=5, firstTauIdx=3, totalLoopNum=7, totalTauNum=8)
    legK = [[1.0, 0, 0, 0, 0, 0, 0], [0, 1.0, 0, 0, 0, 0, 0], [0, 0, 0, 1.0, 0, 0, 0]]
    legK.append([a + c - b for a, b, c in zip(*legK)])
    old = pq.get_ver4I(3)
    new = pq.update_extKT(old, para, legK, para.firstLoopIdx - 1)
    pairs = []
    ids_old = set()

    def walk(a, b):
        ids_old.add(a.id)
        assert b.id not in ids_old
        if not a.subgraphs:
            pairs.append((a.properties, b.properties))
        assert len(a.subgraphs) == len(b.subgraphs) and a.subgraph_factors == b.subgraph_factors
        for x, y in zip(a.subgraphs, b.subgraphs):
            walk(x, y)

    for a, b in zip(old, new):
        walk(a, b)
        assert b.properties.extT == tuple(t + 2 for t in a.properties.extT) and b.properties.para == para
        assert [list(k) for k in b.properties.extK] == [[float(x) for x in k] for k in legK]
    A = np.array([p.extK for p, _ in pairs])                    # [n_leaf, 6]
    Bn = np.array([q.extK for _, q in pairs])                   # [n_leaf, 7]
    assert all(q.extT == tuple(t + 2 for t in p.extT) for p, q in pairs)
    # the legs are unit vectors here, so the map is a placement: find, for every position of the new basis, the component
    # of the old one it copies (by absolute values, the mirror symmetry may flip a leaf's overall sign) ...
    place = {}
    for pnew in range(Bn.shape[1]):
        if not np.any(Bn[:, pnew]):
            continue
        cands = [q for q in range(A.shape[1]) if np.array_equal(np.abs(A[:, q]), np.abs(Bn[:, pnew]))]
        assert len(cands) == 1, (pnew, cands)
        place[pnew] = cands[0]
    assert place == {0: 0, 1: 1, 3: 2, 4: 4, 5: 5, 6: 3}       # legs 1-3 -> their positions; the catalog's inner loop at position 4
    P = np.zeros((A.shape[1], Bn.shape[1]))                    # (the bubble's loop here) moves to 7, the others stay: loops 5..7
    for pnew, q in place.items():
        P[q, pnew] = 1.0
    img = A @ P
    assert np.all(np.all(img == Bn, axis=1) | np.all(-img == Bn, axis=1))       # ... one placement for all leaves, up to that sign
    assert all(tuple(q.extK) == mirror_symmetrize(list(q.extK)) for _, q in pairs)


@pytest.mark.parametrize("name, sizes", [("parquet_sigma5", (274, 11407, 5)), ("parquet_ver4_4", (984, 44854, 180)), ("gv_ver4_4", (1514, 31803, 26))])
def test_larger_graphs_of_the_reference_examples(name, sizes):
    """5-loop Parquet self-energy; the graph of example/benchmark.jl (``Parquet.vertex4``, 4 loops, 180 rows -- its list of
    root indices ``inds`` runs to 178); the graph of example/benchmark_GV.jl:23 (``GV.diagsGV_ver4(4)``)."""
    t = workloads.get(name)
    assert (t.n_leaf, t.n_node, t.n_root) == sizes
    x = np.random.default_rng(2).uniform(0.5, 1.5, size=(2, t.n_leaf))
    assert np.allclose(oracle.eval_static(t, x), oracle.eval_interp(t, x), rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("loops", [1, 2, 3, 4])
def test_vertex_function_sums_agree_with_the_gv_vertex_catalogs(loops):
    """``Parquet.vertex4(DiagPara(type=Ver4Diag, innerLoopNum=n))`` against ``GV.diagsGV_ver4(n)`` (catalogs
    groups_vertex4/Vertex4<n>_0_0.diag: 3, 18, 138, 1 190 Hugenholtz diagrams): with all leaves 1 the UpDown rows sum to the
    catalog's sum of SymFactor * SpinFactor over direct terms -- 2, -9, 40, -168 -- and the UpUp rows (direct + exchange) to 0;
    no sign between them this time.  n = 4 is the graph of example/benchmark.jl against that of example/benchmark_GV.jl."""
    import os
    from feynmandiagram_jl_amd.producers import gv
    want = {1: 2.0, 2: -9.0, 3: 40.0, 4: -168.0}[loops]
    rows = pq.vertex4(DiagPara(type=Ver4Diag, innerLoopNum=loops))
    assert len(rows) == {1: 6, 2: 30, 3: 84, 4: 180}[loops]
    v = all_ones([r["diagram"] for r in rows])
    assert sum(x for x, r in zip(v, rows) if r["response"] == pq.UpDown) == want
    assert sum(x for x, r in zip(v, rows) if r["response"] == pq.UpUp) == 0.0
    if loops == 4:                                   # the shipped table of diagsGV_ver4(4): roots are (UpUp, UpDown) pairs
        t = workloads.get("gv_ver4_4")
        g = oracle.eval_static(t, np.ones((1, t.n_leaf)))[0]
        assert g[1::2].sum() == want and g[0::2].sum() == 0.0
    if os.path.isdir(REF_GV):
        c = gv.parse_vertex4_catalog(f"{REF_GV}/groups_vertex4/Vertex4{loops}_0_0.diag")
        assert _catalog_sums(c) == [0.0, want]


@pytest.mark.parametrize("loops", [1, 2, 3, 4, 5])
def test_polarization_sums_agree_with_the_gv_polarization_catalogs(loops):
    """``Parquet.polarization`` (fermionic, NoHartree) against the GV catalogs groups_charge/Polar<n>_0_0.diag and
    groups_spin/Polar<n>_0_0.diag: with all leaves 1, spin * (UpUp + UpDown) is the charge catalog's sum of
    SymFactor * SpinFactor (-2, 6, -10, -42, 558) and spin * (UpUp - UpDown) the spin catalog's (-2, 6, -18, 46, -66);
    5 loops include the catalog vertex."""
    import os
    from feynmandiagram_jl_amd.producers import gv
    charge = {1: -2.0, 2: 6.0, 3: -10.0, 4: -42.0, 5: 558.0}[loops]
    spin = {1: -2.0, 2: 6.0, 3: -18.0, 4: 46.0, 5: -66.0}[loops]
    rows = pq.polarization(DiagPara(type=pq.PolarDiag, innerLoopNum=loops, filter=(NoHartree,)))
    d = {r["response"]: float(x) for x, r in zip(all_ones([r["diagram"] for r in rows]), rows)}
    uu, ud = d.get(pq.UpUp, 0.0), d.get(pq.UpDown, 0.0)
    assert (2 * (uu + ud), 2 * (uu - ud)) == (charge, spin)
    if os.path.isdir(REF_GV):
        assert all_ones(gv.diagsGV("chargePolar", loops, REF_GV)).tolist() == [charge]
        assert all_ones(gv.diagsGV("spinPolar", loops, REF_GV)).tolist() == [spin]
