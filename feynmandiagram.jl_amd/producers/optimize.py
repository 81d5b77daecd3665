"""``optimize!`` of the reference IR, restated on the host-side graph mirror
(lets the back end accept un-optimized graphs, and is the
only way to obtain *optimized* real graphs here, since Julia is unavailable).

Reference: src/computational_graph/optimize.jl:16-36 (pipeline),
:289-317 (remove_duplicated_leaves!), :345-390 (remove_duplicated_nodes!),
src/computational_graph/transform.jl:354-364 (flatten_chains!), :426-448
(remove_zero_valued_subgraphs!), :472-497 (merge_linear_combination!),
src/computational_graph/abstractgraph.jl:307-349 (isequiv).

All passes mutate the graphs in place, like the reference.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

from ..graph import Graph, Power, Prod, Sum, isleaf, onechild, unary_istrivial

__all__ = ["optimize_", "remove_duplicated_leaves_", "flatten_all_chains_", "merge_all_linear_combinations_",
           "remove_all_zero_valued_subgraphs_", "remove_duplicated_nodes_", "isequiv", "count_operation"]


def _isapprox(a: float, b: float) -> bool:
    return abs(a - b) <= 1.4901161193847656e-08 * max(abs(a), abs(b))


def _props_equal(a, b) -> bool:
    if a is None or b is None:
        return a is b
    return a == b


def isequiv(a: Graph, b: Graph, *skip: str) -> bool:
    """abstractgraph.jl:307-349.  ``skip`` holds field names to ignore."""
    if type(a) is not type(b):
        return False
    if "weight" not in skip and not _isapprox(a.weight, b.weight) and not (a.weight == b.weight):
        return False
    if len(a.subgraph_factors) != len(b.subgraph_factors):
        return False
    if "id" not in skip and a.id != b.id:
        return False
    if "name" not in skip and a.name != b.name:
        return False
    if "orders" not in skip and a.orders != b.orders:
        return False
    if "operator" not in skip and a.operator != b.operator:
        return False
    if "properties" not in skip and not _props_equal(a.properties, b.properties):
        return False
    b_pairs = list(zip(b.subgraphs, b.subgraph_factors))
    for suba, fa in zip(a.subgraphs, a.subgraph_factors):
        for i, (subb, fb) in enumerate(b_pairs):
            if fa == fb and (suba is subb or isequiv(suba, subb, *skip)):
                del b_pairs[i]
                break
        else:
            return False
    return True


def _all_nodes_postorder(graphs: Sequence[Graph]) -> List[Graph]:
    """Unique nodes (by object), children before parents."""
    seen = set()
    out: List[Graph] = []
    for top in graphs:
        stack = [(top, 0)]
        while stack:
            n, i = stack[-1]
            if i == 0 and id(n) in seen:
                stack.pop()
                continue
            if i < len(n.subgraphs):
                stack[-1] = (n, i + 1)
                stack.append((n.subgraphs[i], 0))
            else:
                stack.pop()
                seen.add(id(n))
                out.append(n)
    return out


def _leaf_key(g: Graph):
    p = g.properties
    pk = p.equiv_key() if hasattr(p, "equiv_key") else ("obj", id(p)) if p is not None and not _hashable(p) else p
    return (type(g).__name__, tuple(g.orders), repr(g.operator), pk)


def _hashable(x) -> bool:
    try:
        hash(x)
        return True
    except TypeError:
        return False


def remove_duplicated_leaves_(graphs: Sequence[Graph]) -> Sequence[Graph]:
    """optimize.jl:289-317: leaves sorted by id, the first of every equivalence
    class (isequiv modulo id/name/weight: orders, operator, properties) becomes
    the representative; every parent edge is re-pointed at it."""
    nodes = _all_nodes_postorder(graphs)
    leaves = sorted((n for n in nodes if isleaf(n)), key=lambda g: g.id)
    rep: Dict[object, Graph] = {}
    mapping: Dict[int, Graph] = {}
    for lf in leaves:
        if lf.id in mapping:
            continue
        k = _leaf_key(lf)
        if k not in rep:
            rep[k] = lf
        mapping[lf.id] = rep[k]
    for n in nodes:
        for i, sg in enumerate(n.subgraphs):
            if isleaf(sg):
                n.subgraphs[i] = mapping[sg.id]
    return graphs


def _flatten_chains_(g: Graph) -> None:
    # transform.jl:354-364
    for i, sg in enumerate(g.subgraphs):
        if unary_istrivial(sg) and onechild(sg):
            _flatten_chains_(sg)
            g.subgraph_factors[i] = g.subgraph_factors[i] * sg.subgraph_factors[0]
            g.subgraphs[i] = sg.subgraphs[0]


def flatten_all_chains_(graphs: Sequence[Graph]) -> Sequence[Graph]:
    # optimize.jl:93-101: post-order over every node (idempotent per node)
    for n in _all_nodes_postorder(graphs):
        _flatten_chains_(n)
    return graphs


def _merge_linear_combination_(g: Graph) -> None:
    # transform.jl:472-497
    if not isinstance(g.operator, Sum):
        return
    subg, fac = g.subgraphs, g.subgraph_factors
    added = [False] * len(subg)
    ms: List[Graph] = []
    mf: List[float] = []
    for i in range(len(subg)):
        if added[i]:
            continue
        ms.append(subg[i])
        mf.append(fac[i])
        added[i] = True
        for j in range(i + 1, len(subg)):
            if not added[j] and (subg[i] is subg[j] or isequiv(subg[i], subg[j], "id")):
                added[j] = True
                mf[-1] += fac[j]
    g.subgraphs, g.subgraph_factors = ms, mf


def merge_all_linear_combinations_(graphs: Sequence[Graph]) -> Sequence[Graph]:
    # optimize.jl:186-194 (children first)
    for n in _all_nodes_postorder(graphs):
        _merge_linear_combination_(n)
    return graphs


def _has_zero_subfactors(g: Graph) -> bool:
    # tree_properties.jl:96-114
    if isinstance(g.operator, Sum):
        return all(f == 0 for f in g.subgraph_factors)
    if isinstance(g.operator, Prod):
        return any(f == 0 for f in g.subgraph_factors)
    if isinstance(g.operator, Power):
        return g.subgraph_factors[0] == 0
    return False


def _remove_zero_valued_subgraphs_(g: Graph) -> None:
    # transform.jl:426-448
    if isleaf(g) or (onechild(g) and isleaf(g.subgraphs[0])):
        return
    subg, fac = list(g.subgraphs), list(g.subgraph_factors)
    for i, sg in enumerate(subg):
        if isleaf(sg):
            continue
        if _has_zero_subfactors(sg):
            fac[i] = 0.0
    if isinstance(g.operator, Sum):
        mask = [i for i, f in enumerate(fac) if f != 0] or [0]
    elif isinstance(g.operator, Prod):
        z = [i for i, f in enumerate(fac) if f == 0]
        mask = [z[0]] if z else list(range(len(fac)))
    elif isinstance(g.operator, Power):
        if g.operator.N < 0:
            raise ZeroDivisionError(f"0^{g.operator.N} is illegal!")
        mask = [0]
    else:
        mask = list(range(len(fac)))
    g.subgraphs = [subg[i] for i in mask]
    g.subgraph_factors = [fac[i] for i in mask]


def remove_all_zero_valued_subgraphs_(graphs: Sequence[Graph]) -> Sequence[Graph]:
    # optimize.jl:139-147
    for n in _all_nodes_postorder(graphs):
        _remove_zero_valued_subgraphs_(n)
    return graphs


def remove_duplicated_nodes_(graphs: Sequence[Graph]) -> Sequence[Graph]:
    """optimize.jl:345-390 (level > 0): common-subexpression elimination.  The
    reference compares every node with every unique node (isequiv modulo
    id/name/weight); here equivalence classes are hashed bottom-up on
    (orders, operator, properties, multiset of (child class, factor)), which is
    the same relation."""
    nodes = _all_nodes_postorder(graphs)
    cls: Dict[int, int] = {}          # object id -> equivalence class number
    rep: Dict[object, Graph] = {}     # class key -> representative node
    cls_of_key: Dict[object, int] = {}
    canon: Dict[int, Graph] = {}
    for n in nodes:
        for i, sg in enumerate(n.subgraphs):
            n.subgraphs[i] = canon[id(sg)]
        if isleaf(n):
            k = ("leaf", _leaf_key(n))
        else:
            kids = tuple(sorted((cls[id(sg)], f) for sg, f in zip(n.subgraphs, n.subgraph_factors)))
            p = n.properties
            pk = p.equiv_key() if hasattr(p, "equiv_key") else (p if _hashable(p) else ("obj", id(p)))
            k = (type(n).__name__, tuple(n.orders), repr(n.operator), pk, kids)
        if k not in rep:
            rep[k] = n
            cls_of_key[k] = len(cls_of_key)
        canon[id(n)] = rep[k]
        cls[id(n)] = cls_of_key[k]
    out = [canon[id(g)] for g in graphs]
    if isinstance(graphs, list):
        graphs[:] = out
    return graphs


def optimize_(graphs: Sequence[Graph], level: int = 0) -> Sequence[Graph]:
    """optimize.jl:16-36."""
    if not graphs:
        return None
    if level > 0:
        remove_duplicated_nodes_(graphs)
    else:
        remove_duplicated_leaves_(graphs)
    flatten_all_chains_(graphs)
    merge_all_linear_combinations_(graphs)
    remove_all_zero_valued_subgraphs_(graphs)
    return graphs


def count_operation(graphs: Sequence[Graph]):
    """tree_properties.jl:165-185: [#additions, #multiplications] over unique nodes."""
    adds = mults = 0
    for n in _all_nodes_postorder(graphs):
        k = len(n.subgraphs)
        if k > 0:
            if isinstance(n.operator, Prod):
                mults += k - 1
            elif isinstance(n.operator, Sum):
                adds += k - 1
    return [adds, mults]
