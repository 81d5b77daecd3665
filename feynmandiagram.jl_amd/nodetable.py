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

__all__ = ["NodeTable", "OP_SUM", "OP_PROD", "OP_POWER", "FDG_NO_ROOT",
           "synthetic_parquet_like", "from_program"]


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
                            name=np.array(t.name), leaf_pos=t.leaf_positions().astype(np.uint32))

    @staticmethod
    def load(path) -> "NodeTable":
        z = np.load(path, allow_pickle=False)
        return NodeTable(int(z["n_leaf"]), z["op"], z["power"], z["child_off"], z["child_idx"],
                         z["child_fac"], z["root_slot"], str(z["name"]),
                         z["leaf_pos"] if "leaf_pos" in z.files else None).normalized()

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


def synthetic_parquet_like(n_node: int = 10000, n_leaf: int = 300, n_root: int = 2,
                           seed: int = 20241220, reuse_window: int = 0,
                           name: Optional[str] = None) -> NodeTable:
    """Seeded stand-in for the 4-loop Parquet self-energy graph (SURVEY.md 8d,
    config 3): the real graph needs the Julia front end, which is unavailable.

    Shape follows the survey's spec: Prod:Sum about 2:1, Prod fan-in 2-3, Sum
    fan-in geometric (mean about 3), about 35 % of edges with a factor from
    {-1, -0.5, 0.5, 2, -2}, duplicate-child Prods allowed (cf. g18706 in the
    2-loop fixture), every node reachable from a root (an optimized graph has
    no dead code), ``n_root`` Sum roots.  ``reuse_window`` bounds how far back a
    shared sub-diagram may be re-used (0 = anywhere, the pessimistic case for
    the live set).
    """
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
                        name or f"synthetic_parquet_like_N{n_node}_L{n_leaf}_seed{seed}")
