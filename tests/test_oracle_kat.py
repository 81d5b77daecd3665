"""Pins the oracle (CPU restatement of the reference evaluator) against every
known-answer test the reference's own test-suite holds for this path, and
against the committed golden vectors.  CPU only."""
import json
import math
import os

import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import Compilers, fixtures, workloads
from feynmandiagram_jl_amd.lowering import lower, table_to_Cstr
from feynmandiagram_jl_amd.nodetable import NodeTable

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = {k["name"]: k for k in json.load(open(os.path.join(GOLD, "kat.json")))}


def both(table, leaf):
    leaf = np.atleast_2d(np.asarray(leaf, dtype=np.float64))
    return oracle.eval_static(table, leaf), oracle.eval_interp(table, leaf)


def test_kat_compiler_jl():
    # test/compiler.jl:4-15
    g, leaf, expect = fixtures.kat_compiler_jl()
    t, leafmap, _ = lower([g])
    s, i = both(t, leaf)
    assert s[0, 0] == expect == KAT["compiler_jl"]["expect"][0]
    assert i[0, 0] == expect
    assert len(leafmap) == 2 and t.n_node == 2          # Sum + wrapping Prod(factor=1.5)


def test_kat_evaluation_26_27_702():
    # test/computational_graph.jl:874-887 (exact ==)
    graphs, expect = fixtures.kat_evaluation()
    assert list(expect) == KAT["evaluation_g3_g4_g5"]["expect"]
    for g, e in zip(graphs, expect):
        t, _, _ = lower([g])
        s, i = both(t, np.ones(t.n_leaf))
        assert s[0, 0] == e and i[0, 0] == e
    # all three as one graph set (shared sub-DAG, three roots)
    t, _, _ = lower(list(graphs))
    s, i = both(t, np.ones(t.n_leaf))
    assert s[0].tolist() == list(expect) and i[0].tolist() == list(expect)


@pytest.mark.parametrize("spin,key", [(0.5, "taylor_getdiagram_spin0.5"), (1.0, "front_end_getdiagram_spin1.0")])
def test_kat_getdiagram(spin, key):
    # test/taylor.jl:115-161,202 ; test/front_end.jl:287-309
    root, expect = fixtures.kat_taylor_getdiagram(spin)
    t, _, _ = lower([root])
    s, i = both(t, np.ones(t.n_leaf))
    want = KAT[key]["expect"][0]
    assert math.isclose(s[0, 0], want, rel_tol=1.5e-8)
    assert math.isclose(i[0, 0], want, rel_tol=1.5e-8)
    assert math.isclose(expect, want, rel_tol=1e-15)


def test_sigma2_fixture_all_ones():
    t = workloads.get("sigma2")
    st = t.stats()
    # README.md:59-72 / assets/sigma_o2.svg: 8 leaves, 18 internal (12 Prod, 6 Sum), 37 edges, 2 roots
    assert (st["n_leaf"], st["n_node"], st["n_prod"], st["n_sum"], st["n_edge"], st["n_root"]) == (8, 18, 12, 6, 37, 2)
    assert st["factor_mults"] == 13 and st["flops_alg"] == 32
    s, i = both(t, np.ones(8))
    assert s[0].tolist() == [1.0, -1.0] == KAT["sigma2_all_ones"]["expect"]
    assert i[0].tolist() == [1.0, -1.0]


@pytest.mark.parametrize("name", ["sigma2", "synthetic_small", "sigma4_standin", "sigma4_worstcase", "gv_sigma5", "gv_sigma4_taylor2",
                                  "parquet_sigma4", "parquet_sigma4_taylor2"])
def test_golden_vectors(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    t = NodeTable.load(os.path.join(GOLD, f"{name}.npz"))
    w = workloads.get(name).normalized()
    for a in ("op", "power", "child_off", "child_idx", "child_fac", "root_slot"):
        assert np.array_equal(getattr(t, a), getattr(w, a)), a
    leaf = z["leaf"]
    assert np.array_equal(leaf, oracle.philox_uniform(leaf.shape[0], t.n_leaf, int(z["seed"])))
    assert np.array_equal(oracle.eval_static(t, leaf), z["root_static"])
    assert np.array_equal(oracle.eval_interp(t, leaf), z["root_interp"])
    assert np.array_equal(oracle.eval_static_numpy(t, leaf), z["root_static"])


def test_static_vs_interp_association_differs_but_close():
    t = workloads.get("synthetic_small")
    leaf = oracle.philox_uniform(512, t.n_leaf, 99)
    s, i = both(t, leaf)
    scale = oracle.root_scale(t, leaf)
    assert np.all(np.abs(s - i) <= 1e-12 * np.maximum(1.0, scale))


def test_c_backend_text_compiles_and_matches():
    # the reference's to_Cstr text (static.jl:155-197), gcc -O2 -ffp-contract=off
    for name in ("sigma2", "synthetic_small"):
        t = workloads.get(name)
        cb = oracle.CBaseline(table_to_Cstr(t), t.n_leaf, t.n_root)
        leaf = oracle.philox_uniform(300, t.n_leaf, 5)
        assert np.array_equal(cb(leaf, 1), oracle.eval_static(t, leaf))
        assert np.array_equal(cb(leaf, 3), oracle.eval_static(t, leaf))


@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4", "parquet_sigma4", "gv_sigma5"])
def test_python_backend_text_executes_to_the_oracle_bits(name):
    """A third route to the same bits: the text the reference's compile_Python shape produces (compiler_python.jl:23-47: batched
    `leafVal[:, i]` / `root[:, k]`), executed as it stands by torch on the CPU in Float64 -- elementwise IEEE multiplies and adds in
    Python's left-to-right association, nothing contracted -- against the C oracle.  (Graphs without Power nodes: `**` goes through
    pow().)  (Not so on complex128 tensors: torch's CPU complex product contracts, 1e-17 away from base/complex.jl's formula.)"""
    import torch
    from feynmandiagram_jl_amd.lowering import table_to_python_str
    t = workloads.get(name)
    assert not (np.asarray(t.op) == 2).any()
    ns = {}
    exec(table_to_python_str(t), ns)
    leaf = oracle.philox_uniform(257, t.n_leaf, 19) * 2 - 0.6
    got = ns["eval_graph"](torch.from_numpy(leaf)).numpy()
    assert np.array_equal(got, oracle.eval_static(t, leaf))


def test_kat_taylor_of_gv_sigma_against_the_counterterm_catalogs():
    """test/taylor.jl:97-113 ("Taylor AD of Sigma FeynmanGraph"): with all leaves 1, the Taylor coefficient [GOrder, VerOrder] of
    the 2nd-order GV self-energy (x on fermionic, y on bosonic lines, orders [2, 2]) equals the counter-term catalog
    Sigma2_<VerOrder>_<GOrder>.diag evaluated the same way, for the eight orders the reference tests and both external-time
    groups, exactly (`==`).  The fixture (tests/golden/make_gv_counterterm_kat.py) holds the node table whose 16 roots are
    those coefficients -- built by the restated reader and Taylor pass -- and the 16 numbers computed from the catalog text."""
    t = NodeTable.load(os.path.join(GOLD, "gv_sigma2_counterterm_kat.npz"))
    kat = json.load(open(os.path.join(GOLD, "gv_sigma2_counterterm_kat.json")))
    assert t.n_root == 16 == len(kat["expected"]) and kat["orders"] == [[2, 0, 0], [2, 0, 1], [2, 0, 2], [2, 1, 0], [2, 1, 1], [2, 2, 0], [2, 1, 2], [2, 2, 2]]
    s, i = both(t, np.ones(t.n_leaf))
    assert s[0].tolist() == kat["expected"] == i[0].tolist()
    assert kat["expected"][:2] == [1.0, -1.0] and kat["expected"][-2:] == [18.0, -18.0]


def test_kat_config4_workloads_meet_the_counterterm_catalogs():
    """The same identity on the shipped config-4 graphs: order-k Taylor coefficients in the coupling of the 4- and 5-loop GV
    self-energy, all leaves 1, are the sums of SymFactor * SpinFactor of the catalogs Sigma4_<k>_0.diag / Sigma5_<k>_0.diag
    (numbers from the catalog text, tests/golden/make_gv_counterterm_kat.py: catalog_sums; Sigma5_2_0 does not exist); the
    Parquet form of the 4-loop graph carries the fermionic sign, so its four rows sum to minus the GV totals per order."""
    one = lambda name: oracle.eval_static(workloads.get(name), np.ones((1, workloads.get(name).n_leaf)))[0].tolist()
    assert one("gv_sigma4_taylor2") == [21.0, 3.0, 84.0, 12.0, 210.0, 30.0]            # Sigma4_0_0, Sigma4_1_0, Sigma4_2_0: (dynamic, instant)
    assert one("gv_sigma5_taylor2")[:4] == [-31.0, -77.0, -155.0, -385.0]               # Sigma5_0_0, Sigma5_1_0
    p = np.array(one("parquet_sigma4_taylor2")).reshape(3, 4).sum(axis=1)
    assert p.tolist() == [-(21.0 + 3.0), -(84.0 + 12.0), -(210.0 + 30.0)]


def test_kat_first_derivatives_with_explicit_leaf_vectors():
    """test/computational_graph.jl:930-988: 120, 5, 1 / 570, 3, 1 / 120, 2, 0 / 300, 3840, 480 / 120 on explicit leaf vectors
    (fixtures.kat_first_derivatives: the derivative graphs come from the restated Taylor pass), both evaluators, exact."""
    t, cases = fixtures.kat_first_derivatives()
    assert t.n_leaf == 6 and t.n_root == 5 and len(cases) == 3
    for leaf, want in cases:
        st, it = both(t, leaf)
        for k, w in enumerate(want):
            if w is not None:
                assert st[0, k] == w == it[0, k], (leaf, k, st[0], it[0])


def test_power_nodes():
    g1 = fd.Graph([])
    g2 = fd.Graph([])
    sq = g1 * g1                       # same id => Power(2) (graph.jl:320-321)
    assert isinstance(sq.operator, fd.Power) and sq.operator.N == 2
    cube = fd.multi_product([g1, g1, g1], [2.0, 1.0, 3.0])   # graph.jl:384-386: Power(3), factor 6
    assert cube.operator.N == 3 and cube.subgraph_factors == [6.0]
    p5 = g2 ** 5
    pm = g2 ** -3
    t, _, _ = lower([sq, cube, p5, pm])
    x = np.array([[1.7, 0.9]])
    s, i = both(t, x)
    assert s[0, 0] == 1.7 * 1.7 and s[0, 1] == (1.7 * 1.7 * 1.7) * 6.0
    assert math.isclose(s[0, 2], 0.9 ** 5, rel_tol=4e-16) and math.isclose(s[0, 3], 0.9 ** -3, rel_tol=4e-16)
    assert np.allclose(s, i, rtol=1e-15)
    # product library routine == independent oracle restatement of pow_body
    from feynmandiagram_jl_amd import capi
    for n in (-7, -3, -2, -1, 2, 3, 4, 5, 13, 64):
        for v in (0.3, -1.25, 7.5, 1e-3):
            assert capi.powi(v, n) == oracle.powi(v, n)


def test_root_edge_cases():
    from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT
    a, b = fd.Graph([]), fd.Graph([])
    s = a + b
    # a leaf as root, an interior node as root, an id that is in no graph, a duplicate id
    t, leafmap, ids = lower([s], root=[a.id, s.id, 987654321, s.id])
    assert int(t.root_slot[2]) == FDG_NO_ROOT and int(t.root_slot[3]) == FDG_NO_ROOT   # findfirst (static.jl:112)
    root = np.full((1, 4), -7.0)
    out = oracle.eval_static(t, np.array([[2.0, 3.0]]), root)
    assert out[0].tolist() == [2.0, 5.0, -7.0, -7.0]


def test_green_derive_against_high_precision_vectors():
    """Derivative orders 1..5 of the fermionic Green's function (example/benchmark.jl:93-111).  The derivative
    kernels belong to Lehmann.jl, which is not in the reference checkout, so the oracle restates the definition
    and is pinned by 60-digit mpmath derivatives (tests/golden/make_green_derive.py), not by Lehmann.jl output:
    4 845 points over tau in [-beta, beta] (incl. 0, +-beta), w up to +-700/beta, beta in {1, 3, 25}."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "green_derive.npz"))
    for n in range(1, 6):
        for beta in np.unique(z["beta"]):
            m = (z["order"] == n) & (z["beta"] == beta)
            got = oracle.green_derive(z["tau"][m], z["w"][m], beta, n)
            scale = oracle.green_derive_scale(z["tau"][m], z["w"][m], beta, n)
            assert np.all(np.abs(got - z["value"][m]) <= 1e-12 * np.maximum(scale, 1e-300)), (n, beta)
    # order 1 at tau -> 0+, w = 0: -d/dw [e^{-w tau}/(1+e^{-w beta})] = tau/2 - beta/4
    assert abs(oracle.green_derive(np.array([1e-3]), np.array([0.0]), 2.0, 1)[0] - (1e-3 / 2 - 2.0 / 4)) < 1e-15
    with pytest.raises(NotImplementedError):
        oracle.green_derive(np.array([0.1]), np.array([0.0]), 1.0, 6)
