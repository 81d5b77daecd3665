"""Host-side data model of the computational graph (input of the lowering).

This is the Python mirror of the reference's IR node types, kept to exactly
what the evaluator hot path reads:

* operators ``Sum`` / ``Prod`` / ``Unitary`` / ``Power(N)``
  (reference: src/computational_graph/abstractgraph.jl:3-12),
* ``Graph`` with ``id, name, orders, subgraphs, subgraph_factors, operator,
  weight, properties`` (reference: src/computational_graph/graph.jl:28-75),
* ``FeynmanGraph`` with the same evaluation-relevant fields
  (reference: src/computational_graph/feynmangraph.jl:72-129),
* the arithmetic constructors the reference's own tests use to build their
  known-answer graphs: scalar ``*``, ``linear_combination``, ``+``, ``-``,
  ``multi_product``, ``*``, ``^`` (reference: graph.jl:136-418), including the
  "trivial unary link is merged in place" and "same id => Power(2) / summed
  factor" rules, because those decide the shape (and so the floating-point
  association) of the graphs the evaluator sees.

Nothing here evaluates anything: evaluation lives behind the C ABI
(include/fdg.h).  The diagram front ends, optimizer and Taylor pass of the
reference are out of scope.
"""
from __future__ import annotations

import itertools
import math
from typing import Any, Iterable, Iterator, List, Optional, Sequence

__all__ = [
    "Sum", "Prod", "Unitary", "Power", "AbstractOperator",
    "Graph", "FeynmanGraph", "constant_graph", "linear_combination",
    "multi_product", "PostOrderDFS", "isleaf", "onechild", "unary_istrivial",
    "uid", "reset_uid", "external_vertex", "propagator", "interaction",
]


# --------------------------------------------------------------------------- #
# operators (abstractgraph.jl:3-12)
# --------------------------------------------------------------------------- #
class AbstractOperator:
    """Base of node operators; equality is by type (abstractgraph.jl:15-16)."""

    def __eq__(self, other):
        return type(self) is type(other) and self.__dict__ == other.__dict__

    def __hash__(self):
        return hash((type(self).__name__, tuple(sorted(self.__dict__.items()))))

    def __repr__(self):
        return type(self).__name__


class Sum(AbstractOperator):
    pass


class Prod(AbstractOperator):
    pass


class Unitary(AbstractOperator):
    pass


class Power(AbstractOperator):
    """``Power{N}``; N in {0, 1} is rejected like abstractgraph.jl:8-11."""

    def __init__(self, N: int):
        if not isinstance(N, int) or isinstance(N, bool):
            raise TypeError("Power exponent must be an Int")
        if N in (0, 1):
            raise AssertionError(f"Power{{{N}}} makes no sense.")
        self.N = N

    def __repr__(self):
        return f"Power{{{self.N}}}"


def _as_operator(op) -> AbstractOperator:
    if isinstance(op, AbstractOperator):
        return op
    if isinstance(op, type) and issubclass(op, AbstractOperator):
        return op()
    raise TypeError(f"not an operator: {op!r}")


# --------------------------------------------------------------------------- #
# uid counter (common.jl:1,15-22)
# --------------------------------------------------------------------------- #
_counter = itertools.count(1)


def uid() -> int:
    return next(_counter)


def reset_uid(start: int = 1) -> None:
    """Restart the global id counter (tests use it to get reproducible ids)."""
    global _counter
    _counter = itertools.count(start)


def _isapprox_one(x: float) -> bool:
    # Julia `factor ≈ one(F)`: rtol = sqrt(eps) (graph.jl:69)
    return abs(x - 1.0) <= math.sqrt(2.220446049250313e-16) * max(abs(x), 1.0)


# --------------------------------------------------------------------------- #
# Graph (graph.jl:28-75)
# --------------------------------------------------------------------------- #
class Graph:
    """Computational-graph node; see module docstring for the reference lines.

    A graph-level ``factor`` different from one is lowered *at construction*
    into a wrapping single-child ``Prod`` node carrying the factor on its edge
    (graph.jl:69-73), so the evaluator never sees a node factor.  Because
    Python constructors cannot return a different object, use ``Graph.new``
    (or the module-level arithmetic) when a factor is given; ``Graph(...)``
    with ``factor != 1`` raises to keep the mirror honest.
    """

    __slots__ = ("id", "name", "orders", "subgraphs", "subgraph_factors",
                 "operator", "weight", "properties")

    def __init__(self, subgraphs: Sequence["Graph"] = (), *, subgraph_factors=None,
                 name: str = "", operator=None, orders=None, weight: float = 0.0,
                 properties: Any = None, factor: float = 1.0, _id: Optional[int] = None):
        op = _as_operator(operator) if operator is not None else Sum()
        subgraphs = list(subgraphs)
        if isinstance(op, Power) and len(subgraphs) != 1:
            raise AssertionError("Graph with Power operator must have one and only one subgraph.")
        if isinstance(op, Unitary) and len(subgraphs) != 0:
            raise AssertionError("Graph with Unitary operator must have no subgraphs.")
        if not _isapprox_one(float(factor)):
            raise ValueError("use Graph.new(...) / FeynmanGraph.new(...) when factor != 1 "
                             "(the reference returns a wrapping Prod node, graph.jl:69-73)")
        if subgraph_factors is None:
            subgraph_factors = [1.0] * len(subgraphs)
        subgraph_factors = [float(f) for f in subgraph_factors]
        if len(subgraph_factors) != len(subgraphs):
            raise AssertionError("subgraph_factors and subgraphs differ in length")
        self.id = uid() if _id is None else _id
        self.name = str(name)
        self.orders = list(orders) if orders is not None else [0] * 16
        self.subgraphs: List[Graph] = subgraphs
        self.subgraph_factors: List[float] = subgraph_factors
        self.operator: AbstractOperator = op
        self.weight = weight
        self.properties = properties

    # -- constructor with the reference's `factor` semantics ------------------
    @classmethod
    def new(cls, subgraphs: Sequence["Graph"] = (), *, factor: float = 1.0, **kw):
        g = cls(subgraphs, **kw)
        if _isapprox_one(float(factor)):
            return g
        w = cls([g], subgraph_factors=[float(factor)], operator=Prod(), name=g.name,
                orders=g.orders, properties=g.properties)
        w.weight = g.weight * float(factor)
        return w

    # -- getters used by the back end (compiler.jl:4) -------------------------
    def __repr__(self):
        if not self.subgraphs:
            return f"{self.id}"
        return f"{self.id}={self.operator!r}({','.join(str(s.id) for s in self.subgraphs)})"

    # -- arithmetic (graph.jl:136-418) -----------------------------------------
    def __mul__(self, other):
        if isinstance(other, Graph):
            return multi_product(self, other)
        return _scalar_mul(self, other)

    def __rmul__(self, other):
        return _scalar_mul(self, other)

    def __add__(self, other):
        return linear_combination(self, other, 1.0, 1.0)

    def __sub__(self, other):
        return linear_combination(self, other, 1.0, -1.0)

    def __pow__(self, exponent: int):
        # graph.jl:416-418
        return type(self)([self], operator=Power(int(exponent)),
                          orders=[o * exponent for o in self.orders])


class FeynmanGraph(Graph):
    """Feynman-diagram flavoured node (feynmangraph.jl:72-129).

    Only the evaluation-relevant structure is mirrored; ``properties`` holds an
    opaque record (diagram type, vertices, topology, external indices/legs)
    that ``leafstates`` reads.
    """
    __slots__ = ()


def isleaf(g: Graph) -> bool:  # tree_properties.jl:54
    return len(g.subgraphs) == 0


def onechild(g: Graph) -> bool:  # tree_properties.jl:44
    return len(g.subgraphs) == 1


def unary_istrivial(g: Graph) -> bool:  # abstractgraph.jl:34-35,50
    return isinstance(g.operator, (Sum, Prod))


def _scalar_mul(g1: Graph, c) -> Graph:
    # graph.jl:136-144 / 155-163
    g = type(g1)([g1], subgraph_factors=[float(c)], operator=Prod(), orders=g1.orders)
    if unary_istrivial(g1) and onechild(g1):
        g.subgraph_factors[0] *= g1.subgraph_factors[0]
        g.subgraphs = list(g1.subgraphs)
    return g


def constant_graph(factor: float = 1.0) -> Graph:
    # graph.jl:118-125
    g = Graph([], operator=Unitary(), weight=1.0)
    if _isapprox_one(float(factor)):
        return g
    return g * factor


def _pad_orders(graphs: Sequence[Graph]) -> None:
    n = max(len(g.orders) for g in graphs)
    for g in graphs:
        g.orders = list(g.orders) + [0] * (n - len(g.orders))


def linear_combination(*args, properties=None):
    """``linear_combination(g1, g2, c1=1, c2=1)`` (graph.jl:178-207) or
    ``linear_combination(graphs, constants=ones)`` (graph.jl:228-262)."""
    if isinstance(args[0], Graph):
        g1, g2 = args[0], args[1]
        c1 = float(args[2]) if len(args) > 2 else 1.0
        c2 = float(args[3]) if len(args) > 3 else 1.0
        _pad_orders([g1, g2])
        if g1.orders != g2.orders:
            raise AssertionError("g1 and g2 have different orders.")
        subs, facs = [g1, g2], [c1, c2]
        for i, sg in enumerate((g1, g2)):
            if unary_istrivial(sg) and onechild(sg):
                facs[i] *= sg.subgraph_factors[0]
                subs[i] = sg.subgraphs[0]
        cls = type(g1)
        if subs[0].id == subs[1].id:
            return cls([subs[0]], subgraph_factors=[facs[0] + facs[1]], operator=Sum(),
                       orders=g1.orders, properties=properties)
        return cls(subs, subgraph_factors=facs, operator=Sum(), orders=g1.orders,
                   properties=properties)
    graphs = list(args[0])
    constants = [float(c) for c in (args[1] if len(args) > 1 else [1.0] * len(graphs))]
    _pad_orders(graphs)
    if any(g.orders != graphs[0].orders for g in graphs):
        raise AssertionError("Graphs do not all have the same order.")
    subs, facs = list(graphs), list(constants)
    for i, sg in enumerate(graphs):
        if unary_istrivial(sg) and onechild(sg):
            facs[i] *= sg.subgraph_factors[0]
            subs[i] = sg.subgraphs[0]
    uniq: List[Graph] = []
    ufac: List[float] = []
    for g, f in zip(subs, facs):
        for i, u in enumerate(uniq):
            if u.id == g.id:
                ufac[i] += f
                break
        else:
            uniq.append(g)
            ufac.append(f)
    if not uniq:
        return None
    return type(graphs[0])(uniq, subgraph_factors=ufac, operator=Sum(),
                           orders=graphs[0].orders, properties=properties)


def multi_product(*args, properties=None):
    """``multi_product(g1, g2, c1=1, c2=1)`` (graph.jl:304-331) or
    ``multi_product(graphs, constants=ones)`` (graph.jl:350-401)."""
    if isinstance(args[0], Graph):
        g1, g2 = args[0], args[1]
        c1 = float(args[2]) if len(args) > 2 else 1.0
        c2 = float(args[3]) if len(args) > 3 else 1.0
        subs, facs = [g1, g2], [c1, c2]
        for i, sg in enumerate((g1, g2)):
            if unary_istrivial(sg) and onechild(sg):
                facs[i] *= sg.subgraph_factors[0]
                subs[i] = sg.subgraphs[0]
        cls = type(g1)
        if subs[0].id == subs[1].id:
            return cls([subs[0]], subgraph_factors=[facs[0] * facs[1]], operator=Power(2),
                       orders=[2 * o for o in g1.orders], properties=properties)
        _pad_orders([g1, g2])
        return cls(subs, subgraph_factors=facs, operator=Prod(),
                   orders=[a + b for a, b in zip(g1.orders, g2.orders)], properties=properties)
    graphs = list(args[0])
    constants = [float(c) for c in (args[1] if len(args) > 1 else [1.0] * len(graphs))]
    g1 = graphs[0]
    n = max(len(g.orders) for g in graphs)
    g_orders = [0] * n
    subs, facs = list(graphs), list(constants)
    for i, sg in enumerate(graphs):
        if unary_istrivial(sg) and onechild(sg):
            facs[i] *= sg.subgraph_factors[0]
            subs[i] = sg.subgraphs[0]
        sg.orders = list(sg.orders) + [0] * (n - len(sg.orders))
        g_orders = [a + b for a, b in zip(g_orders, sg.orders)]
    uniq: List[Graph] = []
    ufac: List[float] = []
    reps: List[int] = []
    for g, f in zip(subs, facs):
        for i, u in enumerate(uniq):
            if u.id == g.id:
                ufac[i] *= f
                reps[i] += 1
                break
        else:
            uniq.append(g)
            ufac.append(f)
            reps.append(1)
    if not uniq:
        return None
    cls = type(g1)
    if len(ufac) == 1:
        return cls(uniq, subgraph_factors=ufac, operator=Power(reps[0]), orders=g_orders,
                   properties=properties)
    out = []
    for g, r in zip(uniq, reps):
        out.append(g if r == 1 else cls([g], operator=Power(r), orders=[o * r for o in g1.orders]))
    return cls(out, subgraph_factors=ufac, operator=Prod(), orders=g_orders, properties=properties)


# --------------------------------------------------------------------------- #
# traversal (AbstractTrees.PostOrderDFS over children(g) = subgraphs(g);
# tree_properties.jl:20-22).  Iterative so 10^5-node chains do not recurse.
# --------------------------------------------------------------------------- #
def PostOrderDFS(g: Graph) -> Iterator[Graph]:
    """Children left to right, then the node, over the *tree expansion* of the
    DAG (shared nodes are yielded once per path, as AbstractTrees does)."""
    stack = [(g, 0)]
    while stack:
        node, i = stack[-1]
        if i < len(node.subgraphs):
            stack[-1] = (node, i + 1)
            stack.append((node.subgraphs[i], 0))
        else:
            stack.pop()
            yield node


# --------------------------------------------------------------------------- #
# eval! (src/computational_graph/eval.jl:15-39 for Graph, 42-66 for FeynmanGraph: the same loop).  The reference walks the
# tree, stores `node.weight` for every node and returns the last one; here the leaves get their weights the same way and
# every internal node is a root of ONE device evaluation with eval!'s association (apply, eval.jl:1-13: sum(w f),
# prod(w f), w^N f) -- no arithmetic on the host, no evaluation without the device.
# --------------------------------------------------------------------------- #
def eval_(g: Graph, leafmap=None, leaf=None, *, inherit: bool = False, randseed: int = -1, specialize=False, **kw):
    """``eval!(g, leafmap, leaf; inherit=false, randseed=-1)``.  ``leafmap``: node id -> index into ``leaf`` (0-based here,
    1-based in Julia); empty: every leaf weighs 1.0 (``randseed < 0``) or a uniform random number (drawn per visit from numpy's
    generator seeded with ``randseed`` when it is positive -- Julia's stream is not reproduced); ``inherit``: keep the weights
    the leaves have.  Sets ``weight`` on every node of the graph and returns the root's.  ``specialize``: the back end of the one
    device evaluation -- by default the table interpreter (``fdg_interp``: no JIT for a single sample, any number of roots), or
    "isa" / "auto" / True as in :class:`compilers.GraphFunc`.  (A leaf the walk meets several times keeps
    its last draw and the whole graph is evaluated with it; the reference evaluates each visit's parents with the draw current at that
    moment -- a difference only under ``randseed >= 0`` on shared leaves, where the values are not comparable with Julia's anyway.)"""
    import numpy as np
    from . import compilers
    rng = None
    if randseed >= 0:
        rng = np.random.default_rng(randseed if randseed > 0 else None)
    nodes = list(PostOrderDFS(g))
    for node in nodes:
        if node.subgraphs or inherit:
            continue
        if not leafmap:
            node.weight = 1.0 if rng is None else float(rng.random())
        else:
            node.weight = leaf[leafmap[node.id]]
    inner, seen = [], set()
    for node in nodes:
        if node.subgraphs and node.id not in seen:
            seen.add(node.id)
            inner.append(node)
    if not inner:
        return g.weight
    f, lm = compilers.compile([g], root=[n.id for n in inner], specialize=specialize, association="eval", **kw)
    leaf_val = np.array([float(lm[k + 1].weight) for k in range(len(lm))], dtype=np.float64)
    out = np.zeros(len(inner), dtype=np.float64)
    f(out, leaf_val)
    by_id = {node.id: float(w) for node, w in zip(inner, out)}
    for node in nodes:                      # every visited object, as the reference's loop assigns node.weight on every visit: distinct
        if node.subgraphs:                  # objects that share an id (a structurally duplicated sub-graph) all get the value
            node.weight = by_id[node.id]
    return g.weight


# --------------------------------------------------------------------------- #
# minimal leaf builders for FeynmanGraph KATs (feynmangraph.jl:232-279).  The
# quantum-operator algebra is out of scope: vertices are opaque labels.
# --------------------------------------------------------------------------- #
class _FeynProps:
    __slots__ = ("diagtype", "vertices", "topology", "external_indices", "external_legs")

    def __init__(self, diagtype, vertices, topology=(), external_indices=(), external_legs=()):
        self.diagtype = diagtype
        self.vertices = list(vertices)
        self.topology = list(topology)
        self.external_indices = list(external_indices)
        self.external_legs = list(external_legs)


def external_vertex(ops, name: str = "") -> FeynmanGraph:
    """Leaf of diagram type ExternalVertex (feynmangraph.jl:270-279)."""
    return FeynmanGraph([], operator=Unitary(), name=name,
                        properties=_FeynProps("ExternalVertex", [ops]))


def propagator(ops, name: str = "", factor: float = 1.0, orders=None) -> FeynmanGraph:
    """Leaf of diagram type Propagator (feynmangraph.jl:581-593).  With QuantumOperators (see
    quantum_operators.py) the two operators must be conjugate and the fermionic sign of bringing them to
    correlator order multiplies ``factor``; opaque labels are accepted too (sign then rides in ``factor``)."""
    from .quantum_operators import OperatorProduct, QuantumOperator, correlator_order
    ops = list(ops)
    if ops and all(isinstance(o, QuantumOperator) for o in ops):
        assert len(ops) == 2
        assert ops[0].adjoint.operator == ops[1].operator
        sign, perm = correlator_order(ops)
        props = _FeynProps("Propagator", [OperatorProduct([o]) for o in ops], topology=[[1, 2]],
                           external_indices=perm, external_legs=[True, True])
        return FeynmanGraph.new([], operator=Unitary(), name=name, factor=factor * sign, orders=orders, properties=props)
    return FeynmanGraph.new([], operator=Unitary(), name=name, factor=factor, orders=orders,
                            properties=_FeynProps("Propagator", ops))


def interaction(ops, name: str = "", factor: float = 1.0, orders=None) -> FeynmanGraph:
    """Leaf of diagram type Interaction (feynmangraph.jl:602-613): one bosonic vertex."""
    from .quantum_operators import OperatorProduct, QuantumOperator, isfermionic
    if isinstance(ops, (list, tuple)) and ops and all(isinstance(o, QuantumOperator) for o in ops):
        ops = OperatorProduct(ops)
        assert not isfermionic(ops), "interaction OperatorProduct must be bosonic."
        props = _FeynProps("Interaction", [ops], external_indices=list(range(1, len(ops) + 1)),
                           external_legs=[False] * len(ops))
        return FeynmanGraph.new([], operator=Unitary(), name=name, factor=factor, orders=orders, properties=props)
    return FeynmanGraph.new([], operator=Unitary(), name=name, factor=factor, orders=orders,
                            properties=_FeynProps("Interaction", [ops]))


def diagram_type(g: FeynmanGraph) -> str:
    """feynmangraph.jl:233."""
    return g.properties.diagtype
