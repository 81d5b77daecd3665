"""``FrontEnds.leafstates``: per-leaf tables in ``leafVal`` index order, built
from the ``leafmap`` that ``Compilers.compile`` returns.

Reference: src/frontend/frontends.jl:178-232 (the ``Graph`` method used with the
Parquet / GV graphs: ``leafstates(leaf_maps, maxloopNum)``) and :115-160 (the
``FeynmanGraph`` method with a ``LabelProduct``); type codes
src/frontend/diagram_id.jl:342-354.

The outputs are indexed by the same idx as ``leafVal``: that is why the lowering
reproduces ``to_julia_str``'s leaf numbering exactly.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

from .graph import Graph, isleaf

__all__ = ["leafstates", "index"]


def index(diag_id) -> int:
    """diagram_id.jl:342-354."""
    name = type(diag_id).__name__
    table = {"BareGreenId": 1, "BareInteractionId": 2, "BareGreenNId": 3, "BareHoppingId": 4}
    if name not in table:
        raise NotImplementedError("Not Implemented!")
    return table[name]


def _isapprox_vec(a: Sequence[float], b: Sequence[float]) -> bool:
    # Julia `a ≈ b` on vectors: norm(a - b) <= sqrt(eps) * max(norm(a), norm(b))
    if len(a) != len(b):
        return False
    d = math.sqrt(sum((x - y) ** 2 for x, y in zip(a, b)))
    na = math.sqrt(sum(x * x for x in a))
    nb = math.sqrt(sum(y * y for y in b))
    return d <= 1.4901161193847656e-08 * max(na, nb)


def _leafstates_labelprod(leaf_maps: Sequence[Dict[int, Graph]], labelProd):
    """frontends.jl:115-160: the ``FeynmanGraph`` method.  A leaf is an Interaction vertex (type 0, loop
    index 1, both times from its first operator's label) or a Propagator (type 1 fermionic / 2 bosonic; in
    = second vertex, out = first; loop index = last component of the in-label's index into ``labelProd``).
    Returns ``(leafValue, leafType, leafOrders, leafInTau, leafOutTau, leafLoopIndex)``."""
    from .graph import diagram_type
    from .quantum_operators import isfermionic
    num_g = len(leaf_maps)
    leafType: List[List[int]] = [[] for _ in range(num_g)]
    leafOrders: List[List[List[int]]] = [[] for _ in range(num_g)]
    leafInTau: List[List[int]] = [[] for _ in range(num_g)]
    leafOutTau: List[List[int]] = [[] for _ in range(num_g)]
    leafLoopIndex: List[List[int]] = [[] for _ in range(num_g)]
    leafValue: List[List[float]] = [[] for _ in range(num_g)]
    for ikey, leafmap in enumerate(leaf_maps):
        n = len(leafmap)
        leafValue[ikey] = [1.0] * n
        for idx in range(1, n + 1):
            g = leafmap[idx]
            vertices = g.properties.vertices
            kind = diagram_type(g)
            if kind == "Interaction":
                In = Out = vertices[0][0].label
                leafType[ikey].append(0)
                leafLoopIndex[ikey].append(1)
            elif kind == "Propagator":
                In, Out = vertices[1][0].label, vertices[0][0].label
                leafType[ikey].append(1 if isfermionic(vertices[0]) else 2)
                leafLoopIndex[ikey].append(labelProd.linear_to_index(In)[-1])
            else:
                # the reference falls through both branches and then reads the undefined `In` (UndefVarError)
                raise NameError("In not defined: leaf is neither an Interaction nor a Propagator")
            leafOrders[ikey].append(list(g.orders))
            leafInTau[ikey].append(labelProd[In][0])
            leafOutTau[ikey].append(labelProd[Out][0])
    return leafValue, leafType, leafOrders, leafInTau, leafOutTau, leafLoopIndex


def leafstates(leaf_maps: Sequence[Dict[int, Graph]], maxloopNum):
    """frontends.jl:178-232 (``maxloopNum::Int``; with a ``LabelProduct`` as second argument the
    ``FeynmanGraph`` method of :115-160 is dispatched instead).  Returns
    ``((leafValue, leafType, leafOrders, leafInTau, leafOutTau, leafLoopIndex), loopbasis)``;
    every element of the first tuple is a list with one entry per graph partition."""
    from .labelproduct import LabelProduct
    if isinstance(maxloopNum, LabelProduct):
        return _leafstates_labelprod(leaf_maps, maxloopNum)
    num_g = len(leaf_maps)
    leafType: List[List[int]] = [[] for _ in range(num_g)]
    leafOrders: List[List[List[int]]] = [[] for _ in range(num_g)]
    leafInTau: List[List[int]] = [[] for _ in range(num_g)]
    leafOutTau: List[List[int]] = [[] for _ in range(num_g)]
    leafLoopIndex: List[List[int]] = [[] for _ in range(num_g)]
    leafValue: List[List[float]] = [[] for _ in range(num_g)]
    loopbasis: List[List[float]] = []
    for ikey, leafmap in enumerate(leaf_maps):
        n = len(leafmap)
        leafValue[ikey] = [1.0] * n
        for idx in range(1, n + 1):
            leaf = leafmap[idx]
            assert isleaf(leaf)
            diag_id, leaf_orders = leaf.properties, leaf.orders
            loopmom = [float(x) for x in diag_id.extK]
            assert maxloopNum >= len(loopmom)
            loopmom += [0.0] * (maxloopNum - len(loopmom))
            for bi, basis in enumerate(loopbasis):
                if _isapprox_vec(basis, loopmom):
                    leafLoopIndex[ikey].append(bi + 1)         # 1-based like the reference
                    break
            else:
                loopbasis.append(loopmom)
                leafLoopIndex[ikey].append(len(loopbasis))
            leafInTau[ikey].append(diag_id.extT[0])
            leafOutTau[ikey].append(diag_id.extT[1])
            leafOrders[ikey].append(leaf_orders)
            leafType[ikey].append(index(diag_id))
    return (leafValue, leafType, leafOrders, leafInTau, leafOutTau, leafLoopIndex), loopbasis
