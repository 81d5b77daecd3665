"""Taylor-mode AD on the host-side graph mirror: produces the
enlarged graphs of BASELINE.json config 4 ("Taylor-mode AD counterterms").  Its
*output* is just a bigger DAG of the same node kinds, evaluated by the same kernels.

Reference: src/TaylorSeries/constructors.jl:10-40 (``TaylorSeries{T}``: dict
order -> coefficient), src/TaylorSeries/arithmetic.jl:10-56 (scalar ``*``, ``+``),
:170-191 (truncated product), :282-316 (``^``, power by squaring),
:131-160 (``taylor_binomial``, ``taylor_factorial``), src/TaylorSeries/parameter.jl:
26-35,61-108 (global variable set), src/utility.jl:11-13 (``apply`` on series),
:48-93 (``taylorAD``), :105-135 (``taylorexpansion!`` memoised by node id).

Julia iterates a ``Dict{Vector{Int},T}`` in hash order, which fixes the order in
which products are accumulated into a coefficient; that order is not
reproducible outside Julia, so here coefficients are visited in insertion order.
Values agree up to floating-point reassociation of those sums (the reference's own
tests compare with ``≈``, test/taylor.jl:202-207).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Generic, List, Optional, Sequence, Tuple, TypeVar

from ..graph import Graph, Power, Prod, Sum, isleaf

T = TypeVar("T")
Order = Tuple[int, ...]

__all__ = ["TaylorSeries", "set_variables", "get_orders", "get_numvars", "taylor_factorial", "taylor_binomial",
           "taylorexpansion", "taylorAD", "getcoeff", "coefficient_groups"]

_params = {"orders": [2, 2], "names": ["x1", "x2"]}       # parameter.jl:26


def get_orders() -> List[int]:
    return list(_params["orders"])


def get_numvars() -> int:
    return len(_params["orders"])


class TaylorSeries(Generic[T]):
    def __init__(self, coeffs: Optional[Dict[Order, T]] = None, name: str = ""):
        self.name = name
        self.coeffs: Dict[Order, T] = dict(coeffs or {})

    # arithmetic.jl:10-35
    def _scale(self, c):
        return TaylorSeries({o: c * v for o, v in self.coeffs.items()})

    def __mul__(self, other):
        if isinstance(other, TaylorSeries):
            g: Dict[Order, T] = {}
            orders = get_orders()
            for o1, c1 in self.coeffs.items():                    # arithmetic.jl:170-191
                for o2, c2 in other.coeffs.items():
                    o = tuple(a + b for a, b in zip(o1, o2))
                    if all(x <= m for x, m in zip(o, orders)):
                        g[o] = (g[o] + c1 * c2) if o in g else (c1 * c2)
            return TaylorSeries(g)
        return self._scale(other)

    __rmul__ = _scale

    def __add__(self, other):
        if isinstance(other, TaylorSeries):                        # arithmetic.jl:44-56
            g = dict(self.coeffs)
            for o, c in other.coeffs.items():
                g[o] = (g[o] + c) if o in g else c
            return TaylorSeries(g)
        g = dict(self.coeffs)                                      # constant (arithmetic.jl:91-102)
        z = (0,) * get_numvars()
        g[z] = (g[z] + other) if z in g else other
        return TaylorSeries(g)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-1 * other)

    def __rsub__(self, other):
        return other + (-1 * self)

    def __pow__(self, p: int):
        # arithmetic.jl:282-316
        if p == 1:
            return TaylorSeries(self.coeffs)
        if p == 0:
            raise NotImplementedError("one(x) needs a coefficient type")
        if p == 2:
            return self * self
        if p < 0:
            raise ValueError("DomainError")
        x = self
        t = (p & -p).bit_length()          # trailing_zeros(p) + 1
        p >>= t
        while t > 1:
            t -= 1
            x = x * x
        y = x
        while p > 0:
            t = (p & -p).bit_length()
            p >>= t
            while t > 0:
                t -= 1
                x = x * x
            y = y * x
        return y


def set_variables(names, orders: Optional[Sequence[int]] = None, dtype=float) -> List[TaylorSeries]:
    """parameter.jl:61-108: returns one series per variable."""
    if isinstance(names, str):
        names = names.split()
    orders = list(orders) if orders is not None else get_orders()
    if len(names) < 1:
        raise ValueError("Number of variables must be at least 1")
    if len(orders) != len(names):
        raise AssertionError("Input orders should have same length as number of variables.")
    _params["orders"], _params["names"] = orders, list(names)
    out = []
    for i in range(len(names)):
        v = [0] * len(names)
        v[i] = 1
        out.append(TaylorSeries({tuple(v): dtype(1)}))
    return out


def getcoeff(g: TaylorSeries, order: Sequence[int]):
    return g.coeffs.get(tuple(order))


def taylor_factorial(o: Sequence[int]) -> int:
    r = 1
    for x in o:
        r *= math.factorial(x)
    return r


def taylor_binomial(o1: Sequence[int], o2: Sequence[int]) -> int:
    r = 1
    for a, b in zip(o1, o2):
        if a + b:
            r *= math.comb(a + b, a)
    return r


def _apply(graph: Graph, series: List[TaylorSeries], factors: List[float]) -> TaylorSeries:
    # utility.jl:11-13
    terms = [d * f for d, f in zip(series, factors)]
    op = graph.operator
    if isinstance(op, Sum):
        acc = terms[0]
        for t in terms[1:]:
            acc = acc + t
        return acc
    if isinstance(op, Prod):
        acc = terms[0]
        for t in terms[1:]:
            acc = acc * t
        return acc
    if isinstance(op, Power):
        return (series[0] ** op.N) * factors[0]
    raise NotImplementedError(repr(op))


def taylorexpansion(graph, var_dependence: Optional[Dict[int, List[bool]]] = None,
                    to_coeff_map: Optional[Dict[int, TaylorSeries]] = None):
    """``taylorexpansion!`` (utility.jl:105-135); a list of graphs gives a list of series (:221-230).
    Iterative post-order so deep graphs do not recurse."""
    var_dependence = var_dependence or {}
    m = to_coeff_map if to_coeff_map is not None else {}
    if isinstance(graph, (list, tuple)):
        return [taylorexpansion(g, var_dependence, m)[0] for g in graph], m
    nv = get_numvars()
    stack = [(graph, 0)]
    while stack:
        g, i = stack[-1]
        if g.id in m:
            stack.pop()
            continue
        if isleaf(g):
            stack.pop()
            var = var_dependence.get(g.id, [False] * nv)
            ranges = [range(0, get_orders()[k] + 1) if var[k] else range(0, 1) for k in range(nv)]
            res: Dict[Order, Graph] = {}

            def rec(k, cur):                      # Iterators.product: first index fastest
                if k < 0:
                    o = tuple(cur)
                    res[o] = g if sum(o) == 0 else type(g)([], operator=Sum(), properties=g.properties, orders=list(o))
                    return
                for v in ranges[k]:
                    cur[k] = v
                    rec(k - 1, cur)
            rec(nv - 1, [0] * nv)
            m[g.id] = TaylorSeries(res)
            continue
        if i < len(g.subgraphs):
            stack[-1] = (g, i + 1)
            stack.append((g.subgraphs[i], 0))
            continue
        stack.pop()
        ts = _apply(g, [m[s.id] for s in g.subgraphs], g.subgraph_factors)
        for c in ts.coeffs.values():
            c.properties = g.properties
        m[g.id] = ts
    return m[graph.id], m


def coefficient_groups(to_coeff_map: Dict[int, TaylorSeries]) -> Dict[int, int]:
    """node id of every Taylor coefficient graph -> id of the original node it was derived from
    (the scheduling hint of fdg_graph_set_schedule_groups)."""
    out: Dict[int, int] = {}
    for oid, ts in to_coeff_map.items():
        for c in ts.coeffs.values():
            out.setdefault(c.id, oid)
    return out


def taylorAD(graphs: Sequence[Graph], deriv_orders: Sequence[int], leaf_dep_funcs: Sequence[Callable],
             dict_graphs: Optional[Dict[Order, List[Graph]]] = None, groups: Optional[Dict[int, int]] = None
             ) -> Dict[Order, List[Graph]]:
    """utility.jl:48-93.  ``groups`` (optional, filled in place) receives ``coefficient_groups``."""
    if len(deriv_orders) != len(leaf_dep_funcs):
        raise AssertionError("Lengths of deriv_orders and properties_deps must be equal.")
    names = []
    for i in range(len(deriv_orders)):
        names.append(chr(ord("a") + i) if i < 26 else names[i - 26] + chr(ord("a") + (i % 26)))
    set_variables(names, orders=list(deriv_orders))
    dep: Dict[int, List[bool]] = {}
    seen = set()
    for g in graphs:
        stack = [g]
        while stack:
            n = stack.pop()
            if id(n) in seen:
                continue
            seen.add(id(n))
            if isleaf(n):
                dep.setdefault(n.id, [bool(f(n.properties)) for f in leaf_dep_funcs])
            else:
                stack.extend(n.subgraphs)
    series, cmap = taylorexpansion(list(graphs), dep)
    if groups is not None:
        groups.update(coefficient_groups(cmap))
    out = dict_graphs if dict_graphs is not None else {}
    for ts in series:
        for o, g in ts.coeffs.items():
            out.setdefault(o, []).append(g)
    return out
