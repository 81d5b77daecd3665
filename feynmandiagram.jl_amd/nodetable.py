"""Flat, topologically sorted node table: the interchange format between the
host-side lowering and the C ABI (``fdg_graph_desc`` in include/fdg.h).

Value index space: leaves ``0 .. L-1`` (in ``leafVal`` order, i.e. the order of
first visit of the reference's ``to_julia_str`` traversal,
src/backend/static.jl:104,115-120), then internal nodes ``L .. L+N-1`` in the
order their statements are emitted (post-order, first visit wins,
static.jl:121-125).  ``child_idx`` points into that space, so every child of a
node has a smaller index than the node: the table is a straight-line program.

``root_slot[k]`` is the value index written to ``root[k]`` (static.jl:111-114,
126-128); ``FDG_NO_ROOT`` marks a requested root id that no graph contains (the
reference then simply never assigns ``root[k]``).
"""
from __future__ import annotations

import io
import random
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

OP_SUM, OP_PROD, OP_POWER = 0, 1, 2
FDG_NO_ROOT = 0xFFFFFFFF

__all__ = ["NodeTable", "OP_SUM", "OP_PROD", "OP_POWER", "FDG_NO_ROOT", "complex_to_real",
           "synthetic_parquet_like", "from_program", "postorder_renumber"]


@dataclass
class NodeTable:
    n_leaf: int
    op: np.ndarray          # uint8  [N]
    power: np.ndarray       # int32  [N]  (exponent for OP_POWER, else 0)
    child_off: np.ndarray   # uint32 [N+1]
    child_idx: np.ndarray   # uint32 [E]
    child_fac: np.ndarray   # float64[E]
    root_slot: np.ndarray   # uint32 [R]
    name: str = ""
    # text-emitter detail only (no effect on arithmetic): number of internal-node
    # statements that precede leaf k's load statement in the reference's output.
    leaf_pos: Optional[np.ndarray] = None   # uint32 [L], non-decreasing
    # optional scheduling hint (fdg_graph_set_schedule_groups): nodes with equal tags are evaluated together
    sched_group: Optional[np.ndarray] = None   # uint32 [N]

    # ------------------------------------------------------------------ #
    @property
    def n_node(self) -> int:
        return int(self.op.shape[0])

    @property
    def n_edge(self) -> int:
        return int(self.child_idx.shape[0])

    @property
    def n_root(self) -> int:
        return int(self.root_slot.shape[0])

    def normalized(self) -> "NodeTable":
        """dtype/contiguity normalisation (what the ctypes layer hands over)."""
        return NodeTable(
            int(self.n_leaf),
            np.ascontiguousarray(self.op, dtype=np.uint8),
            np.ascontiguousarray(self.power, dtype=np.int32),
            np.ascontiguousarray(self.child_off, dtype=np.uint32),
            np.ascontiguousarray(self.child_idx, dtype=np.uint32),
            np.ascontiguousarray(self.child_fac, dtype=np.float64),
            np.ascontiguousarray(self.root_slot, dtype=np.uint32),
            self.name,
            None if self.leaf_pos is None else np.ascontiguousarray(self.leaf_pos, dtype=np.uint32),
            None if self.sched_group is None else np.ascontiguousarray(self.sched_group, dtype=np.uint32),
        )

    def leaf_positions(self) -> np.ndarray:
        """``leaf_pos`` or, when the table did not come from ``lower``, the latest
        placement that keeps leaves in index order and ahead of their first use."""
        if self.leaf_pos is not None:
            return self.leaf_pos.astype(np.int64)
        L, N = self.n_leaf, self.n_node
        first = np.full(L + 1, N, dtype=np.int64)
        k = np.diff(self.child_off.astype(np.int64))
        owner = np.repeat(np.arange(N, dtype=np.int64), k)
        ci = self.child_idx.astype(np.int64)
        m = ci < L
        np.minimum.at(first, ci[m], owner[m])
        return np.minimum.accumulate(first[:L][::-1])[::-1] if L else first[:0]

    def validate(self) -> None:
        """Structural checks; the C ABI repeats them (fdg_graph_create)."""
        N, L = self.n_node, self.n_leaf
        if L < 0:
            raise ValueError("n_leaf < 0")
        if self.power.shape[0] != N or self.child_off.shape[0] != N + 1:
            raise ValueError("array length mismatch")
        if N and int(self.child_off[0]) != 0:
            raise ValueError("child_off[0] != 0")
        if int(self.child_off[-1]) != self.n_edge or self.child_fac.shape[0] != self.n_edge:
            raise ValueError("child_off[N] != n_edge")
        off = self.child_off.astype(np.int64)
        k = np.diff(off)
        if (k < 1).any():
            raise ValueError("internal node without children")
        if (self.op > OP_POWER).any():
            raise ValueError("unknown operator code")  # static.jl:6-11
        pw = self.op == OP_POWER
        if (k[pw] != 1).any():
            raise ValueError("Power node must have exactly one child")  # graph.jl:61-62
        if np.isin(self.power[pw], (0, 1)).any():
            raise ValueError("Power{0}/Power{1} make no sense")  # abstractgraph.jl:9
        owner = np.repeat(np.arange(N, dtype=np.int64), k) + L
        if (self.child_idx.astype(np.int64) >= owner).any():
            raise ValueError("child index not smaller than its node (not topologically sorted)")
        rs = self.root_slot.astype(np.int64)
        if ((rs >= L + N) & (rs != FDG_NO_ROOT)).any():
            raise ValueError("root_slot out of range")

    # -- op counts in the reference's own terms (tree_properties.jl:165-185) -- #
    def stats(self) -> dict:
        k = np.diff(self.child_off.astype(np.int64))
        nonunit = int((self.child_fac != 1.0).sum())
        is_sum, is_prod, is_pow = (self.op == OP_SUM), (self.op == OP_PROD), (self.op == OP_POWER)
        adds = int((k[is_sum] - 1).sum())
        mults = int((k[is_prod] - 1).sum())
        pw = np.abs(self.power[is_pow].astype(np.int64))
        pow_mults = int(np.where(pw <= 3, pw - 1, 2 * np.ceil(np.log2(np.maximum(pw, 2)))).sum())
        L, R = self.n_leaf, self.n_root
        return dict(n_leaf=L, n_node=self.n_node, n_edge=self.n_edge, n_root=R,
                    n_sum=int(is_sum.sum()), n_prod=int(is_prod.sum()), n_power=int(is_pow.sum()),
                    adds=adds, mults=mults, factor_mults=nonunit, pow_mults=pow_mults,
                    flops_alg=adds + mults + nonunit + pow_mults,
                    bytes_alg=8 * (L + R), bytes_alg_accumulate=8 * L)

    # -- (de)serialisation: one .npz, used for golden fixtures ------------- #
    def save(self, path) -> None:
        t = self.normalized()
        np.savez_compressed(path, n_leaf=np.int64(t.n_leaf), op=t.op, power=t.power,
                            child_off=t.child_off, child_idx=t.child_idx,
                            child_fac=t.child_fac, root_slot=t.root_slot,
                            name=np.array(t.name), leaf_pos=t.leaf_positions().astype(np.uint32),
                            **({} if t.sched_group is None else {"sched_group": t.sched_group}))

    @staticmethod
    def load(path) -> "NodeTable":
        z = np.load(path, allow_pickle=False)
        return NodeTable(int(z["n_leaf"]), z["op"], z["power"], z["child_off"], z["child_idx"],
                         z["child_fac"], z["root_slot"], str(z["name"]),
                         z["leaf_pos"] if "leaf_pos" in z.files else None,
                         z["sched_group"] if "sched_group" in z.files else None).normalized()

    def children(self, n: int) -> List[Tuple[int, float]]:
        a, b = int(self.child_off[n]), int(self.child_off[n + 1])
        return [(int(self.child_idx[e]), float(self.child_fac[e])) for e in range(a, b)]


def from_program(n_leaf: int, nodes: Sequence[Tuple[int, int, Sequence[Tuple[int, float]]]],
                 roots: Sequence[int], name: str = "") -> NodeTable:
    """Build a table from ``[(op, power, [(child_value_index, factor), ...]), ...]``."""
    op = np.array([n[0] for n in nodes], dtype=np.uint8)
    power = np.array([n[1] for n in nodes], dtype=np.int32)
    off = np.zeros(len(nodes) + 1, dtype=np.uint32)
    idx: List[int] = []
    fac: List[float] = []
    for i, n in enumerate(nodes):
        for c, f in n[2]:
            idx.append(c)
            fac.append(f)
        off[i + 1] = len(idx)
    t = NodeTable(n_leaf, op, power, off, np.array(idx, dtype=np.uint32),
                  np.array(fac, dtype=np.float64), np.array(list(roots), dtype=np.uint32), name)
    t.validate()
    return t


def complex_to_real(t: NodeTable) -> NodeTable:
    """The graph of ``t`` on complex values spelled out on their real and imaginary parts -- the Float64 table whose leaves are
    ``re_0, im_0, re_1, im_1, ...`` (a row of a row-major ComplexF64 ``[B, L]`` matrix read as ``2 L`` doubles) and whose roots are
    ``re, im`` of the original roots.  Every operation is the one Julia performs on ``Complex{Float64}`` values, with the
    association the evaluator's own folds keep (base/complex.jl; base/operators.jl):

    * ``z * f`` (Float64 factor): ``(re * f, im * f)`` -- one-child nodes ``(g * f)``;
    * ``z + w``: componentwise, so a Sum node becomes two Sum nodes with the same children order and factors;
    * ``z * w = (zr wr - zi wi, zr wi + zi wr)``: four products, ``P1 + P2 * -1.0`` (``x * -1.0`` is ``-x`` exactly) and ``P3 + P4``;
      an n-ary Prod is the left fold of these with the factors applied where the text applies them;
    * ``z^2 = z * z``, ``z^3 = (z * z) * z`` (Base.literal_pow); other exponents are not covered.

    The Float64 evaluator then gives the bits of the generic function on ComplexF64 arguments (checked against
    ``oracle.eval_static_typed`` and against the per-type device kernel in tests/test_typed.py)."""
    t = t.normalized()
    L = t.n_leaf
    off, idx, fac = t.child_off, t.child_idx, t.child_fac
    nodes: List[Tuple[int, int, List[Tuple[int, float]]]] = []
    val: List[Optional[Tuple[int, int]]] = [(2 * l, 2 * l + 1) for l in range(L)] + [None] * t.n_node

    def add(op, children):
        nodes.append((op, 0, children))
        return 2 * L + len(nodes) - 1

    def scale(z, f):
        return z if f == 1.0 else (add(OP_SUM, [(z[0], f)]), add(OP_SUM, [(z[1], f)]))

    def cmul(z, w):
        p1, p2 = add(OP_PROD, [(z[0], 1.0), (w[0], 1.0)]), add(OP_PROD, [(z[1], 1.0), (w[1], 1.0)])
        re = add(OP_SUM, [(p1, 1.0), (p2, -1.0)])
        p3, p4 = add(OP_PROD, [(z[0], 1.0), (w[1], 1.0)]), add(OP_PROD, [(z[1], 1.0), (w[0], 1.0)])
        return re, add(OP_SUM, [(p3, 1.0), (p4, 1.0)])

    for n in range(t.n_node):
        a, b = int(off[n]), int(off[n + 1])
        ch = [(val[int(idx[e])], float(fac[e])) for e in range(a, b)]
        o = int(t.op[n])
        if o == OP_SUM:
            z = (add(OP_SUM, [(c[0], f) for c, f in ch]), add(OP_SUM, [(c[1], f) for c, f in ch]))
        elif o == OP_PROD:
            z = scale(ch[0][0], ch[0][1])
            for c, f in ch[1:]:
                z = scale(cmul(z, c), f)
        elif o == OP_POWER and int(t.power[n]) in (2, 3):
            x = ch[0][0]
            z = cmul(x, x)
            if int(t.power[n]) == 3:
                z = cmul(z, x)
            z = scale(z, ch[0][1])
        else:
            raise NotImplementedError("complex_to_real covers Sum, Prod, Power{2}, Power{3}")
        val[L + n] = z
    roots: List[int] = []
    for k in range(t.n_root):
        s_ = int(t.root_slot[k])
        if s_ == FDG_NO_ROOT:
            roots += [FDG_NO_ROOT, FDG_NO_ROOT]
        else:
            roots += [val[s_][0], val[s_][1]]
    return from_program(2 * L, nodes, roots, name=(t.name or "graph") + ":complex_as_real")


def synthetic_parquet_like(n_node: int = 10000, n_leaf: int = 300, n_root: int = 2,
                           seed: int = 20241220, structure: str = "recursive",
                           name: Optional[str] = None) -> NodeTable:
    """Seeded stand-in for the 4-loop Parquet self-energy graph (SURVEY.md 8d,
    config 3): the real graph needs the Julia front end, which is unavailable.

    Shape follows the survey's spec: about ``n_node`` internal nodes, ``n_leaf``
    leaves, ``n_root`` Sum roots, Prod:Sum about 2:1, Prod fan-in 2-3, Sum fan-in
    geometric (mean about 3), about 35 % of edges with a factor from
    {-1, -0.5, 0.5, 2, -2}, duplicate-child Prods allowed (cf. g18706 in the
    2-loop fixture), every node reachable from a root.

    ``structure`` decides how sub-diagrams are shared, which the spec leaves open:

    * ``"recursive"`` (default) mimics the parquet recursion the reference's
      builder runs (src/frontend/parquet/vertex4.jl:66-99,125-201, sigma.jl:20-136):
      vertex functions Gamma(l, variant) of loop order l = 0..3 are sums over
      channels and loop splits l1 + l2 = l - 1 of
      ``Gamma(l1, v1) * (G*G) * Gamma(l2, v2)``; each Gamma(l, variant) is built
      once and shared by every parent that needs it (low orders are shared
      widely, high orders by few parents); Sigma = sum of ``G * Gamma(3, v)``.
    * ``"random"``: every operand is drawn uniformly from all earlier nodes -- no
      locality at all, about a quarter of all values alive at once.  Kept as the
      worst case for the register allocator (workload ``sigma4_worstcase``).
    """
    if structure == "random":
        return _synthetic_random(n_node, n_leaf, n_root, seed, name)
    rng = random.Random(seed)
    facs = (-1.0, -0.5, 0.5, 2.0, -2.0)
    L = n_leaf
    n_g = (L * 5) // 6                 # propagator leaves G; the rest are interaction leaves V
    nodes: List[Tuple[int, int, List[Tuple[int, float]]]] = []

    def fac() -> float:
        return rng.choice(facs) if rng.random() < 0.35 else 1.0

    def add(op: int, ch) -> int:
        nodes.append((op, 0, list(ch)))
        return L + len(nodes) - 1

    gg_cache = {}

    def gg() -> int:                   # a shared pair of propagators G_a * G_b
        a, b = rng.randrange(n_g), rng.randrange(n_g)
        if (a, b) not in gg_cache:
            gg_cache[(a, b)] = add(OP_PROD, [(a, 1.0), (b, fac())])
        return gg_cache[(a, b)]

    # variants per loop order, sized so that the total lands near n_node
    scale = max(n_node, 200) / 10000.0
    n_var = [max(4, int(round(x * scale))) for x in (32, 128, 500, 1250)]
    gamma: List[List[int]] = [[] for _ in range(4)]
    # order 0: direct - exchange combinations of bare interaction leaves
    for _v in range(n_var[0]):
        k = 2
        while rng.random() < 0.4 and k < 4:
            k += 1
        gamma[0].append(add(OP_SUM, [(n_g + rng.randrange(L - n_g), fac()) for _ in range(k)]))
    for l in range(1, 4):
        for _v in range(n_var[l]):
            terms = []
            n_terms = 2
            while rng.random() < 0.5 and n_terms < 12:      # geometric, mean about 3
                n_terms += 1
            for _t in range(n_terms):
                l1 = rng.randrange(l)
                l2 = l - 1 - l1
                left = rng.choice(gamma[l1])
                right = rng.choice(gamma[l2])               # left == right happens: duplicate-child Prod
                if _t and rng.random() < 0.3 and l >= 2:
                    terms.append((rng.choice(gamma[l - 1]), fac()))     # lower-order piece re-used as is
                elif rng.random() < 0.5:
                    terms.append((add(OP_PROD, [(left, fac()), (gg(), fac()), (right, fac())]), fac()))
                else:
                    terms.append((add(OP_PROD, [(left, fac()), (right, fac())]), fac()))
            gamma[l].append(add(OP_SUM, terms))
    # Sigma roots: sum over G * Gamma(3, v), every top-level vertex used once
    tops = list(gamma[3])
    rng.shuffle(tops)
    used = set()
    for lst in nodes:
        for c, _f in lst[2]:
            used.add(c)
    # lower-order vertex functions nobody picked hang directly under a root too
    extra = [v for l in range(3) for v in gamma[l] if v not in used]
    tops += extra
    roots = []
    for r in range(n_root):
        grp = tops[r::n_root]
        terms = [(add(OP_PROD, [(rng.randrange(n_g), 1.0), (v, fac())]), fac()) for v in grp]
        roots.append(add(OP_SUM, terms))
    t = from_program(L, nodes, roots, "")
    t = postorder_renumber(t)          # statement order of to_julia_str (static.jl:98-133)
    t.name = name or f"synthetic_parquet_recursive_N{t.n_node}_L{t.n_leaf}_seed{seed}"
    return t


def postorder_renumber(t: NodeTable) -> NodeTable:
    """Renumber leaves and nodes in the order the reference's code generator
    visits them: post-order DFS from the roots in order, children left to right,
    first visit wins; unreachable nodes are dropped, unreachable leaves keep
    their relative order after the visited ones."""
    L, N = t.n_leaf, t.n_node
    new_leaf: dict = {}
    new_node: dict = {}
    order: List[int] = []
    leaf_pos: List[int] = []
    for top in [int(r) for r in t.root_slot if int(r) != FDG_NO_ROOT]:
        stack = [(top, 0)]
        while stack:
            v, i = stack[-1]
            if v < L:
                stack.pop()
                if v not in new_leaf:
                    new_leaf[v] = len(new_leaf)
                    leaf_pos.append(len(order))
                continue
            n = v - L
            if i == 0 and n in new_node:
                stack.pop()
                continue
            a, b = int(t.child_off[n]), int(t.child_off[n + 1])
            if i < b - a:
                stack[-1] = (v, i + 1)
                stack.append((int(t.child_idx[a + i]), 0))
            else:
                stack.pop()
                new_node[n] = len(order)
                order.append(n)
    for v in range(L):
        if v not in new_leaf:
            new_leaf[v] = len(new_leaf)
            leaf_pos.append(len(order))
    nodes = []
    for n in order:
        ch = [((new_leaf[c] if c < L else L + new_node[c - L]), f) for c, f in t.children(n)]
        nodes.append((int(t.op[n]), int(t.power[n]), ch))
    roots = []
    for r in t.root_slot:
        r = int(r)
        roots.append(FDG_NO_ROOT if r == FDG_NO_ROOT else (new_leaf[r] if r < L else L + new_node[r - L]))
    out = from_program(L, nodes, roots, t.name)
    out.leaf_pos = np.array(leaf_pos, dtype=np.uint32)
    return out


def _synthetic_random(n_node: int, n_leaf: int, n_root: int, seed: int, name: Optional[str]) -> NodeTable:
    reuse_window = 0
    rng = random.Random(seed)
    facs = (-1.0, -0.5, 0.5, 2.0, -2.0)
    L = n_leaf
    nodes = []
    unused: List[int] = []
    nvals = L
    for _ in range(n_node - n_root):
        isprod = rng.random() < 2.0 / 3.0
        if isprod:
            k = rng.choice((2, 2, 3))
        else:
            k = 2
            while rng.random() < 0.5 and k < 12:
                k += 1
        ch = []
        for _j in range(k):
            r = rng.random()
            if unused and r < min(0.9, len(unused) / 300.0):
                c = unused.pop(rng.randrange(len(unused)))
            elif r < 0.6 or nvals == L:
                c = rng.randrange(L)
            else:
                lo = L if reuse_window <= 0 else max(L, nvals - reuse_window)
                c = lo + rng.randrange(nvals - lo)
            f = rng.choice(facs) if rng.random() < 0.35 else 1.0
            ch.append((c, f))
        nodes.append((OP_PROD if isprod else OP_SUM, 0, ch))
        unused.append(nvals)
        nvals += 1
    rng.shuffle(unused)
    roots = []
    for r in range(n_root):
        grp = unused[r::n_root] or [rng.randrange(nvals)]
        ch = [(c, rng.choice(facs) if rng.random() < 0.35 else 1.0) for c in grp]
        nodes.append((OP_SUM, 0, ch))
        roots.append(nvals)
        nvals += 1
    return from_program(L, nodes, roots,
                        name or f"synthetic_random_dag_N{n_node}_L{n_leaf}_seed{seed}")
