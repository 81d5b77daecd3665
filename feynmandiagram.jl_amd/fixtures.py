"""Graphs the reference's own tests and assets define for the evaluator path,
re-built with the host-side mirror so parity tests read like the reference's.

* ``sigma2_graphs``: the optimized 2-loop Parquet self-energy (configs 1-2 of
  BASELINE.json).  The Julia front end cannot run here; the structure is
  transcribed from the reference's own ``compile_dot`` rendering of exactly
  this graph, assets/sigma_o2.svg (README.md:59-72,138-145) -- node ids, child
  order and factors as in SURVEY.md Appendix A.
* ``kat_*``: the known-answer graphs of test/compiler.jl:4-15,
  test/computational_graph.jl:874-887 and test/taylor.jl:115-161.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

from .graph import FeynmanGraph, Graph, Prod, Sum, external_vertex, reset_uid


def sigma2_graphs() -> Tuple[List[Graph], Dict[str, Graph]]:
    """Returns ``([SigmaIns, SigmaDyn], nodes_by_name)``; roots g18721 / g18722."""
    n: Dict[int, Graph] = {}

    def leaf(i, kind):
        n[i] = Graph([], _id=i, name=kind)
        return n[i]

    def node(i, op, ch, fac=None):
        n[i] = Graph([n[c] for c in ch], subgraph_factors=fac, operator=op, _id=i)
        return n[i]

    leaf(18636, "G"); leaf(18643, "G"); leaf(18637, "V"); leaf(18650, "G")
    leaf(18630, "V"); leaf(18676, "V"); leaf(18708, "G"); leaf(18709, "G")
    node(18644, Prod(), [18643, 18637], [1.0, -1.0])
    node(18651, Prod(), [18644, 18650])
    node(18652, Prod(), [18636, 18651])
    node(18653, Prod(), [18652, 18630], [1.0, -1.0])
    node(18721, Sum(), [18653])
    node(18682, Sum(), [18630, 18676], [-1.0, 1.0])
    node(18691, Sum(), [18630, 18676], [-1.0, 1.0])
    node(18694, Prod(), [18682, 18691])
    node(18706, Prod(), [18630, 18630], [-1.0, -1.0])
    node(18714, Sum(), [18694, 18706], [-1.0, -1.0])
    node(18715, Prod(), [18714, 18708, 18709])
    node(18717, Prod(), [18636, 18715])
    node(18698, Prod(), [18630, 18682], [-1.0, 1.0])
    node(18702, Prod(), [18630, 18691], [-1.0, 1.0])
    node(18710, Sum(), [18698, 18702], [-1.0, -1.0])
    node(18711, Prod(), [18708, 18709, 18710])
    node(18719, Prod(), [18636, 18711])
    node(18722, Sum(), [18717, 18719], [1.0, -0.5])
    return [n[18721], n[18722]], {f"g{k}": v for k, v in n.items()}


# The straight-line program the reference's to_julia_str emits for that graph
# (statement order of static.jl:98-133).
SIGMA2_JULIA_BODY = """\
    g18636 = leafVal[1]
    g18643 = leafVal[2]
    g18637 = leafVal[3]
    g18644 = (g18643 * g18637 * -1.0)
    g18650 = leafVal[4]
    g18651 = (g18644 * g18650)
    g18652 = (g18636 * g18651)
    g18630 = leafVal[5]
    g18653 = (g18652 * g18630 * -1.0)
    g18721 = (g18653)
    root[1] = g18721
    g18676 = leafVal[6]
    g18682 = (g18630 * -1.0 + g18676)
    g18691 = (g18630 * -1.0 + g18676)
    g18694 = (g18682 * g18691)
    g18706 = (g18630 * -1.0 * g18630 * -1.0)
    g18714 = (g18694 * -1.0 + g18706 * -1.0)
    g18708 = leafVal[7]
    g18709 = leafVal[8]
    g18715 = (g18714 * g18708 * g18709)
    g18717 = (g18636 * g18715)
    g18698 = (g18630 * -1.0 * g18682)
    g18702 = (g18630 * -1.0 * g18691)
    g18710 = (g18698 * -1.0 + g18702 * -1.0)
    g18711 = (g18708 * g18709 * g18710)
    g18719 = (g18636 * g18711)
    g18722 = (g18717 + g18719 * -0.5)
    root[2] = g18722
"""


def kat_compiler_jl():
    """test/compiler.jl:4-15: ``g = FeynmanGraph([ev1, ev2]; factor=1.5)``;
    ``eval_graph!([0.0], [1.0, 2.0]) == 4.5`` and the call returns that value."""
    subgraphs = [external_vertex("f+(1)f-(2)"), external_vertex("f+(3)f-(4)")]
    g = FeynmanGraph.new(subgraphs, factor=1.5)
    return g, [1.0, 2.0], 4.5


def kat_evaluation():
    """test/computational_graph.jl:874-887: all leaves 1 => 26, 27, 27*26."""
    g1 = Graph([])
    g2 = Graph.new([], factor=2)
    g3 = 2 * (3 * g1 + 5 * g2)
    g4 = g1 + 2 * (3 * g1 + 5 * g2)
    g5 = g4 * g3
    return (g3, g4, g5), (26.0, 27.0, 27.0 * 26.0)


def kat_taylor_getdiagram(spin: float = 2.0, D: int = 3):
    """test/taylor.jl:115-161 ``getdiagram``: all leaves 1 => (spin-2)/(2pi)^D
    (taylor.jl:202 with spin=0.5; test/front_end.jl:290,307 with spin=1.0)."""
    g = [Graph([], name="G") for _ in range(2)]
    vd = [Graph([], name="Vd") for _ in range(2)]
    ve = [Graph([], name="Ve") for _ in range(2)]
    ggn = Graph([g[0], g[1]], operator=Prod())
    vdd = Graph.new([vd[0], vd[1]], operator=Prod(), factor=spin)
    vde = Graph.new([vd[0], ve[1]], operator=Prod(), factor=-1.0)
    ved = Graph.new([ve[0], vd[1]], operator=Prod(), factor=-1.0)
    vsum = Graph([vdd, vde, ved], operator=Sum())
    root = Graph.new([vsum, ggn], operator=Prod(), factor=1 / (2 * math.pi) ** D, name="root")
    return root, (spin - 2.0) / (2 * math.pi) ** D


def kat_first_derivatives(with_graphs: bool = False):
    """test/computational_graph.jl:930-988 (the "forwardAD_root!" set): F3 = g1 + g2, F2 = 2 g1 + g3 + 3 F3 with g3 = 2 * leaf,
    F1 = (3 g1) F2 F3, F0 = F1 F3, F0' = F1 + F3; the reference evaluates the first-derivative graphs of F1, F2, F3, F0, F0' on
    leaf vectors [g1, g2, g3, dg1, dg2, dg3] and holds the values below (exact ==).  The legacy graph AD that builds those graphs
    is out of scope; the same directional derivatives are the order-1 Taylor coefficients of the restated Taylor pass with every
    leaf depending on the variable.  Returns ``(table, [(leaf vector in leafVal order, [expected root or None, ...]), ...])``:
    known answers of the reference on leaf vectors that are not all ones, through the lowering's leaf numbering."""
    import numpy as np
    from .lowering import lower
    from .graph import linear_combination
    from .producers import taylor
    reset_uid()
    g1, g2, g3 = Graph([]), Graph([]), Graph.new([], factor=2.0)
    F3 = g1 + g2
    F2 = linear_combination([g1, g3, F3], [2, 1, 3])
    F1 = Graph([g1, F2, F3], operator=Prod(), subgraph_factors=[3.0, 1.0, 1.0])
    F0 = F1 * F3
    F0r = F1 + F3
    taylor.set_variables("x", orders=[1])
    leaves = [g1, g2, g3.subgraphs[0]]                 # eldest(g3): the leaf under the wrapping Prod that carries the factor
    series, cmap = taylor.taylorexpansion([F1, F2, F3, F0, F0r], {l.id: [True] for l in leaves})
    table, leafmap, _ = lower([s.coeffs[(1,)] for s in series], name="kat_first_derivatives")
    where = {id(cmap[l.id].coeffs[(o,)]): k + 3 * o for k, l in enumerate(leaves) for o in (0, 1)}
    cases = []
    for values, want in (((1.0, 1.0, 1.0, 1.0, 0.0, 0.0), [120.0, 5.0, 1.0, 300.0, None]),
                         ((5.0, -1.0, 2.0, 0.0, 1.0, 0.0), [570.0, 3.0, 1.0, 3840.0, None]),
                         ((5.0, -1.0, 2.0, 0.0, 0.0, 1.0), [120.0, 2.0, 0.0, 480.0, 120.0])):
        v = np.zeros(table.n_leaf)
        for idx, obj in leafmap.items():
            v[idx - 1] = values[where[id(obj)]]
        cases.append((v, want))
    if with_graphs:      # the call shape of the reference's test: eval!(graph, leafmap, leaf) with leafmap = node id -> index into leaf
        graphs = [s.coeffs[(1,)] for s in series]
        id_leafmap = {obj.id: where[id(obj)] for obj in leafmap.values()}
        vectors = [(1.0, 1.0, 1.0, 1.0, 0.0, 0.0), (5.0, -1.0, 2.0, 0.0, 1.0, 0.0), (5.0, -1.0, 2.0, 0.0, 0.0, 1.0)]
        return table, cases, graphs, id_leafmap, vectors
    return table, cases
