"""the reference's SECOND evaluator, the interpreter `eval!` (src/computational_graph/eval.jl:1-3,15-39),
which its examples and almost all of its tests call (example/benchmark.jl:84-86), rounds differently from the generated
code: every operand is scaled by its factor before it enters the fold -- Prod = (w1 f1) * (w2 f2) * ... against the
generated ((w1 f1) * w2) * f2 ...  A handle with FDG_ASSOC_INTERP (GraphFunc(..., association="eval")) reproduces it bit for
bit on every back end.  Oracle: oracle.eval_interp (oracle/fdg_oracle.c: eval_one_interp).

CPU: the allocated programs (one-wave and cooperative) replayed with IEEE operations; GPU: the three back ends, the layouts,
fused accumulation, the reference's known answers."""
import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, fixtures, workloads
from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT, OP_POWER, OP_PROD, OP_SUM, from_program
from feynmandiagram_jl_amd.lowering import lower
from test_next_rows import replay, replay_coop
from test_random_graphs import random_table, same


def scaled_products_table():
    """Products whose later operands carry factors that are not +-2^k: where the two evaluators must part."""
    nodes = [
        (OP_PROD, 0, [(0, 1.0), (1, 3.0), (2, 1.0 / 3.0)]),            # ((a * b) * 3) * c * (1/3)   vs   a * (b * 3) * (c * (1/3))
        (OP_PROD, 0, [(3, 0.7), (0, -7.5), (4, 1e-3), (1, 1.0)]),
        (OP_SUM, 0, [(5, 1.0), (6, -1.0), (2, 0.3)]),
        (OP_POWER, 2, [(7, 1.7)]),
        (OP_PROD, 0, [(8, 1.0), (7, 1.1), (5, -0.9)]),
    ]
    return from_program(5, nodes, [9, 7, 5], "scaled_products")


def eval_kat():
    return lower(list(fixtures.kat_evaluation()[0]))[0]


def compiler_kat():
    return lower([fixtures.kat_compiler_jl()[0]])[0]


def test_the_two_evaluators_part_on_scaled_products_and_nowhere_on_the_baseline_graphs():
    t = scaled_products_table()
    leaf = oracle.philox_uniform(257, t.n_leaf, 3) * 4 - 2
    a, b = oracle.eval_static(t, leaf), oracle.eval_interp(t, leaf)
    assert (a.view(np.int64) != b.view(np.int64)).any()
    assert np.allclose(a, b, rtol=1e-10, atol=1e-12)        # (sums that cancel: a few ulps of the terms, not of the result)
    # the graphs of BASELINE.json: after optimize! a Prod carries factors +-1 (or +-2^k) beyond its first operand, where
    # (acc * w) * f == acc * (w * f) exactly -- the two evaluators agree bit for bit there (a finding, asserted so that it stays one)
    for name in ("sigma2", "parquet_sigma4", "gv_sigma5", "gv_sigma4_taylor2", "parquet_sigma4_taylor2"):
        w = workloads.get(name)
        x = oracle.philox_uniform(33, w.n_leaf, 5)
        assert np.array_equal(oracle.eval_static(w, x).view(np.int64), oracle.eval_interp(w, x).view(np.int64)), name


def test_known_answers_of_the_interpreter():
    """test/computational_graph.jl:874-887 is an eval! test: 26, 27, 702 exact; test/compiler.jl:4-15's 4.5."""
    t = eval_kat()
    assert oracle.eval_interp(t, np.ones((1, t.n_leaf))).tolist() == [[26.0, 27.0, 702.0]]
    t = compiler_kat()
    assert oracle.eval_interp(t, np.array([[1.0, 2.0]])).tolist() == [[4.5]]


@pytest.mark.parametrize("seed", list(range(24)))
def test_interp_association_program_replays_to_eval_interp(libfdg, seed):
    t = random_table(seed) if seed else scaled_products_table()
    h = capi.GraphHandle(t)
    h.set_association(capi.FDG_ASSOC_INTERP)
    leaf = oracle.philox_uniform(9, t.n_leaf, seed + 100) * 4 - 2
    want = oracle.eval_interp(t, leaf)
    live = t.root_slot != FDG_NO_ROOT
    for budget in (dict(), dict(n_reg=7, n_lds=2, n_acc=2, lookahead_leaf=9, vn_window=1)):
        ops, nr, nl, nm = h.opt_program(**budget)
        got = replay(ops, nr, nl, nm, h.last_n_acc, leaf, t.n_root)
        assert same(got[:, live], want[:, live]), (seed, budget)
    # and the default handle still gives the generated code's bits on the same table
    h0 = capi.GraphHandle(t)
    ops, nr, nl, nm = h0.opt_program()
    assert same(replay(ops, nr, nl, nm, h0.last_n_acc, leaf, t.n_root)[:, live], oracle.eval_static(t, leaf)[:, live])


@pytest.mark.parametrize("name", ["sigma4_standin", "synthetic_small"])
def test_interp_association_cooperative_program(libfdg, monkeypatch, name, fdgopt):
    """the stand-ins have thousands of scaled operands inside products (+-0.5, +-2: exact) -- the cooperative schedule keeps
    the interpreter's association too"""
    fdgopt.set("FDG_COOP_WAVES", "4")
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    h.set_association(capi.FDG_ASSOC_INTERP)
    progs, info = h.coop_program()
    leaf = oracle.philox_uniform(5, t.n_leaf, 79)
    assert np.array_equal(replay_coop(progs, info, leaf, t.n_root), oracle.eval_interp(t, leaf))


def test_association_is_chosen_before_specialisation(libfdg, tmp_path):
    t = workloads.get("sigma2")
    h = capi.GraphHandle(t)
    h.set_association(capi.FDG_ASSOC_INTERP)
    h.set_association(capi.FDG_ASSOC_INTERP)          # idempotent
    h.specialize(str(tmp_path), capi.FDG_SPEC_ISA)
    with pytest.raises(capi.FdgError) as e:
        h.set_association(capi.FDG_ASSOC_STATIC)
    assert e.value.code == capi.FDG_E_INVALID
    with pytest.raises(capi.FdgError):
        capi.GraphHandle(t).set_association(7)
    with pytest.raises(ValueError):
        fd.compile_table(t, association="julia")


def test_interp_association_in_the_emitted_hip_source(libfdg, monkeypatch, fdgopt):
    """statement-order text (FDG_HIP_TABLE_ORDER=1): a scaled operand is parenthesised before it is folded"""
    fdgopt.set("FDG_HIP_TABLE_ORDER", "1")
    t = scaled_products_table()
    h = capi.GraphHandle(t)
    static_src = h.emit_source()
    h.set_association(capi.FDG_ASSOC_INTERP)
    interp_src = h.emit_source()
    assert static_src != interp_src
    assert "(g1 * 0x1.8p+1)" in interp_src and "(g1 * 0x1.8p+1)" not in static_src


# ---------------------------------------------------------------- device ----------------------------------------------------------------
GPU_GRAPHS = ["scaled_products", "sigma2", "parquet_sigma4", "gv_sigma5", "gv_sigma4_taylor2", "eval_kat", "compiler_kat", "random_3", "random_17"]


def _table(name):
    if name == "scaled_products": return scaled_products_table()
    if name == "eval_kat": return eval_kat()
    if name == "compiler_kat": return compiler_kat()
    if name.startswith("random_"): return random_table(int(name.split("_")[1]))
    return workloads.get(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_GRAPHS)
def test_three_back_ends_reproduce_eval_interp(libfdg, cuda, name):
    import torch
    t = _table(name)
    B = 4099
    h_leaf = oracle.philox_uniform(B, t.n_leaf, 21) * (4 if name.startswith(("random", "scaled")) else 1) - (2 if name.startswith(("random", "scaled")) else 0)
    if name.endswith("_kat"):
        h_leaf[0] = 1.0
        if name == "compiler_kat": h_leaf[0] = [1.0, 2.0]
    want = oracle.eval_interp(t, h_leaf, np.full((B, t.n_root), 9.0))
    live = t.root_slot != FDG_NO_ROOT
    for spec in ("isa", True, False):
        f = fd.compile_table(t, specialize=spec, association="eval")
        for layout in ("row", "col"):
            leaf = torch.from_numpy(h_leaf if layout == "row" else np.asfortranarray(h_leaf)).to(cuda)
            if layout == "col":
                leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
            root = torch.full((B, t.n_root), 9.0, dtype=torch.float64, device=cuda)
            f(root, leaf)
            torch.cuda.synchronize()
            got = root.cpu().numpy()
            assert same(got[:, live], want[:, live]), (name, spec, layout, f.kernel_info()["last_kernel"])
            assert (got[:, ~live] == 9.0).all()
        if spec == "isa":      # fused accumulation keeps the association too (sum over samples: 1e-12 of the term scale)
            acc = f.accumulate(leaf).cpu().numpy()
            ref = want[:, live].sum(axis=0)
            scale = np.abs(want[:, live]).sum(axis=0) + 1.0
            ok = np.isfinite(ref)
            assert np.all(np.abs(acc[live][ok] - ref[ok]) <= 1e-12 * scale[ok]), name
    if name == "eval_kat":
        assert want[0].tolist() == [26.0, 27.0, 702.0]


@pytest.mark.gpu
def test_interp_and_static_handles_differ_on_device_where_the_oracles_do(libfdg, cuda):
    import torch
    t = scaled_products_table()
    h_leaf = oracle.philox_uniform(1000, t.n_leaf, 3) * 4 - 2
    leaf = torch.from_numpy(h_leaf).to(cuda)
    a = fd.compile_table(t, specialize="isa")(None, leaf).cpu().numpy()
    b = fd.compile_table(t, specialize="isa", association="eval")(None, leaf).cpu().numpy()
    assert np.array_equal(a, oracle.eval_static(t, h_leaf)) and np.array_equal(b, oracle.eval_interp(t, h_leaf))
    assert (a != b).any()


# ---- `eval!` itself: ComputationalGraphs.eval_ (host mirror of eval.jl:15-39) -------------------------------------------------
def _graph_with_scaled_products():
    """A small graph in which eval!'s rounding differs from the generated function's (factors that are not powers of two on later
    operands of products), built with the host mirror's own constructors."""
    from feynmandiagram_jl_amd.graph import Graph, Prod, Sum, Power
    a, b, c, d = (Graph([]) for _ in range(4))
    p1 = Graph([a, b, c], subgraph_factors=[1.0, 3.0, 1.0 / 3.0], operator=Prod())
    p2 = Graph([d, a, p1], subgraph_factors=[0.7, -7.5, 1e-3], operator=Prod())
    s = Graph([p1, p2, c], subgraph_factors=[1.0, -1.0, 0.3], operator=Sum())
    q = Graph([s], subgraph_factors=[1.7], operator=Power(2))
    top = Graph([q, s, p1], subgraph_factors=[1.0, 1.1, -0.9], operator=Prod())
    return top, [a, b, c, d]


def test_eval_bang_mirror_has_no_cpu_fallback(libfdg):
    """Without a gfx950 device the call fails loudly (FDG_E_NO_DEVICE); the leaves have their weights by then, as in the reference's loop."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    (g3, _, _), _ = fixtures.kat_evaluation()
    with pytest.raises(capi.FdgError) as e:
        fd.eval_(g3)
    assert e.value.code == capi.FDG_E_NO_DEVICE


@pytest.mark.gpu
def test_eval_bang_mirror_on_device(libfdg, cuda):
    """test/computational_graph.jl:874-887 through the mirror of the function the reference's test calls: eval!(g3) == 26,
    eval!(g4) == 27, eval!(g5) == 27 * 26; every node's weight is set; leafmap / leaf, inherit and randseed as eval.jl:15-39."""
    from feynmandiagram_jl_amd.graph import PostOrderDFS
    (g3, g4, g5), want = fixtures.kat_evaluation()
    assert (fd.eval_(g3), fd.eval_(g4), fd.eval_(g5)) == want
    assert (fd.eval_(g3, specialize="isa"), fd.eval_(g4, specialize=True), fd.eval_(g5, specialize="auto")) == want      # every back end
    assert g5.weight == 702.0 and all(n.weight == 1.0 for n in PostOrderDFS(g5) if not n.subgraphs)
    # a graph on which eval! and the generated function round differently: every node's weight against the oracle's interpreter
    top, leaves = _graph_with_scaled_products()
    vals = [0.37, -1.9, 2.4, 0.051]
    leafmap = {l.id: k for k, l in enumerate(leaves)}
    got = fd.eval_(top, leafmap, vals)
    assert fd.eval_(top, leafmap, vals, specialize="isa") == got == fd.eval_(top, leafmap, vals, specialize=True)
    inner, seen = [], set()
    for n in PostOrderDFS(top):
        if n.subgraphs and n.id not in seen:
            seen.add(n.id); inner.append(n)
    t, lm, _ = lower([top], root=[n.id for n in inner])
    x = np.array([[vals[leafmap[lm[k + 1].id]] for k in range(len(lm))]])
    ref = oracle.eval_interp(t, x)[0]
    assert [n.weight for n in inner] == ref.tolist() and got == ref[-1] == top.weight
    assert not np.array_equal(ref, oracle.eval_static(t, x)[0])           # (the case separates the two evaluators)
    assert [l.weight for l in leaves] == vals
    # inherit=true keeps the leaves' weights; randseed > 0 draws them (reproducibly here, not Julia's stream)
    for l, v in zip(leaves, (1.5, 2.0, -0.25, 4.0)):
        l.weight = v
    r1 = fd.eval_(top, inherit=True)
    assert [l.weight for l in leaves] == [1.5, 2.0, -0.25, 4.0] and r1 == oracle.eval_interp(t, np.array([[lm[k + 1].weight for k in range(len(lm))]]))[0][-1]
    r2, w2 = fd.eval_(top, randseed=7), [l.weight for l in leaves]
    assert fd.eval_(top, randseed=7) == r2 and [l.weight for l in leaves] == w2 and all(0.0 <= w < 1.0 for w in w2)
    # a bare leaf: its weight, no device call
    assert fd.eval_(leaves[0], inherit=True) == leaves[0].weight
    # test/computational_graph.jl:471-491: the value does not change under optimize!, leaves drawn with the same seed
    from feynmandiagram_jl_amd.graph import Graph, Prod
    from feynmandiagram_jl_amd.producers import optimize
    from test_next_rows import O
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
    optimize.optimize_([h])
    assert fd.eval_(h, randseed=2) == pytest.approx(fd.eval_(_h, randseed=2), rel=1e-14)
    assert fd.eval_(h) == fd.eval_(_h) == (-28 + 3) * 2 + 3
    # distinct objects that carry the same id (a structurally duplicated sub-graph) all get their weight, as eval! assigns
    # node.weight on every visit
    import copy
    la, lb = Graph([]), Graph([])
    sm = Graph([la, lb], subgraph_factors=[2, 3])
    sm2 = copy.copy(sm)
    assert sm2 is not sm and sm2.id == sm.id
    top2 = Graph([sm, sm2], operator=Prod())
    la.weight, lb.weight = 1.5, -0.5
    assert fd.eval_(top2, inherit=True) == 1.5 * 1.5 and sm.weight == sm2.weight == 1.5



@pytest.mark.gpu
def test_eval_bang_mirror_with_leafmap_and_leaf_vectors(libfdg, cuda):
    """test/computational_graph.jl:930-988 in the reference's own call shape, ``eval!(dual[k], leafmap, leaf)``: 120, 5, 1, 300 /
    570, 3, 1, 3840 / 120, 2, 0, 480, 120 (exact ==) on leaf vectors that are not all ones.  (The derivative graphs come from the
    restated Taylor pass: fixtures.kat_first_derivatives.)"""
    _, cases, graphs, leafmap, vectors = fixtures.kat_first_derivatives(with_graphs=True)
    for (_, want), leaf in zip(cases, vectors):
        for g, w in zip(graphs, want):
            if w is not None:
                assert fd.eval_(g, leafmap, list(leaf)) == w

