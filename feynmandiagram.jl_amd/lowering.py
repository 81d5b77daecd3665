"""Lowering of a list of graphs to the flat node table, in exactly the
statement order of the reference's code generators, plus the reference's three
text emitters restated over that table.

Reference being followed:
* traversal / numbering / root mapping: src/backend/static.jl:98-133
  (``to_julia_str``), identical in ``to_Cstr`` (static.jl:155-197) and
  ``to_python_str`` (src/backend/compiler_python.jl:9-52);
* per-node expression text: ``to_static`` (static.jl:13-46).

The reference walks the *tree expansion* of the DAG with ``PostOrderDFS`` and
skips nodes whose id it has already seen.  A DFS that does not descend into an
already-emitted node yields the same first-visit order (every descendant of an
emitted node was emitted before it), in O(N + E) instead of O(N^2).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .graph import Graph, Power, Prod, Sum, Unitary
from .nodetable import FDG_NO_ROOT, OP_POWER, OP_PROD, OP_SUM, NodeTable

__all__ = ["lower", "to_julia_str", "to_Cstr", "to_python_str", "table_to_Cstr",
           "table_to_julia_str", "table_to_python_str"]


def _opcode(g: Graph) -> Tuple[int, int]:
    op = g.operator
    if isinstance(op, Sum):
        return OP_SUM, 0
    if isinstance(op, Prod):
        return OP_PROD, 0
    if isinstance(op, Power):
        return OP_POWER, op.N
    # static.jl:6-11: any other operator on a node with children is an error
    raise NotImplementedError(
        f"Static representation for computational graph nodes with operator {op!r} not yet implemented!")


def lower(graphs: Sequence[Graph], root: Optional[Sequence[int]] = None,
          name: str = "", groups: Optional[Dict[int, int]] = None) -> Tuple[NodeTable, Dict[int, Graph], Dict[int, int]]:
    """Returns ``(table, leafmap, value_index_of_id)``.  ``groups`` (node id -> tag) becomes the
    table's scheduling hint; untagged nodes are their own group.

    ``leafmap`` maps the 1-based ``leafVal`` index to the leaf graph object,
    exactly the second return value of ``to_julia_str`` (static.jl:104,117-119).
    """
    graphs = list(graphs)
    root_ids = [g.id for g in graphs] if root is None else [int(r) for r in root]
    first_pos: Dict[int, int] = {}
    for pos, rid in enumerate(root_ids):          # findfirst (static.jl:112)
        first_pos.setdefault(rid, pos)

    leaf_index: Dict[int, int] = {}               # id -> 0-based leaf index
    node_index: Dict[int, int] = {}               # id -> 0-based internal index
    leafmap: Dict[int, Graph] = {}
    order: List[Graph] = []                       # internal nodes in emission order
    leaf_pos: List[int] = []                      # nodes emitted before each leaf load

    for top in graphs:
        stack: List[Tuple[Graph, int]] = [(top, 0)]
        while stack:
            node, i = stack[-1]
            gid = node.id
            if i == 0 and (gid in leaf_index or gid in node_index):
                stack.pop()                        # `continue` at static.jl:116,122
                continue
            if i < len(node.subgraphs):
                stack[-1] = (node, i + 1)
                stack.append((node.subgraphs[i], 0))
                continue
            stack.pop()
            if not node.subgraphs:
                leaf_index[gid] = len(leaf_index)
                leafmap[len(leaf_index)] = node    # 1-based key
                leaf_pos.append(len(order))
            else:
                _opcode(node)                      # reject unknown operators now
                node_index[gid] = len(order)
                order.append(node)

    L = len(leaf_index)

    def vidx(gid: int) -> int:
        return leaf_index[gid] if gid in leaf_index else L + node_index[gid]

    op = np.zeros(len(order), dtype=np.uint8)
    power = np.zeros(len(order), dtype=np.int32)
    off = np.zeros(len(order) + 1, dtype=np.uint32)
    idx: List[int] = []
    fac: List[float] = []
    for n, g in enumerate(order):
        op[n], power[n] = _opcode(g)
        for sg, f in zip(g.subgraphs, g.subgraph_factors):
            idx.append(vidx(sg.id))
            fac.append(float(f))
        off[n + 1] = len(idx)

    root_slot = np.full(len(root_ids), FDG_NO_ROOT, dtype=np.uint32)
    for rid, pos in first_pos.items():
        if rid in leaf_index or rid in node_index:
            root_slot[pos] = vidx(rid)

    table = NodeTable(L, op, power, off, np.array(idx, dtype=np.uint32),
                      np.array(fac, dtype=np.float64), root_slot, name,
                      np.array(leaf_pos, dtype=np.uint32))
    if groups:
        tags = {}
        table.sched_group = np.array([tags.setdefault(("g", groups[g.id]) if g.id in groups else ("n", g.id), len(tags))
                                      for g in order], dtype=np.uint32)
    table.validate()
    ids = {gid: i for gid, i in leaf_index.items()}
    ids.update({gid: L + i for gid, i in node_index.items()})
    return table, leafmap, ids


# --------------------------------------------------------------------------- #
# text emitters over the table (names g<value index + 1> unless ids are given)
# --------------------------------------------------------------------------- #
def _fstr(f: float) -> str:
    # Julia interpolates Float64 with its shortest round-trip repr; Python's
    # repr is the same digits (exponent spelling may differ; both parse exactly).
    if f != f or f in (float("inf"), float("-inf")):
        raise ValueError("non-finite subgraph factor")
    return repr(float(f))


def _expr(table: NodeTable, n: int, names: Sequence[str], lang: str) -> str:
    ch = table.children(n)
    o = int(table.op[n])
    if o == OP_POWER:
        c, f = ch[0]
        fs = "" if f == 1 else f" * {_fstr(f)}"
        N = int(table.power[n])
        if lang == "c":
            return f"pow({names[c]}, {N}){fs}"           # static.jl:38-39
        return f"(({names[c]}){'^' if lang == 'julia' else '**'}{N}{fs})"   # static.jl:45
    terms = [names[c] + ("" if f == 1 else f" * {_fstr(f)}") for c, f in ch]
    if len(terms) == 1:
        return f"({terms[0]})"                            # static.jl:14-16,24-26
    return "(" + (" + " if o == OP_SUM else " * ").join(terms) + ")"


def _names(table: NodeTable, ids: Optional[Dict[int, int]]) -> List[str]:
    n = table.n_leaf + table.n_node
    if ids is None:
        return [f"g{i + 1}" for i in range(n)]
    inv = {v: k for k, v in ids.items()}
    return [f"g{inv[i]}" for i in range(n)]


def _emit(table: NodeTable, ids, lang: str):
    """Statement list in reference order: leaf loads in index order, each at the
    point of the walk's first visit (``table.leaf_positions()``), internal nodes
    in index order (static.jl:115-125)."""
    L, N = table.n_leaf, table.n_node
    names = _names(table, ids)
    roots_of: Dict[int, int] = {}
    for k, s in enumerate(table.root_slot):
        if int(s) != FDG_NO_ROOT:
            roots_of.setdefault(int(s), k)
    pos = table.leaf_positions()
    stmts: List[Tuple[str, int]] = []     # ("leaf"|"node", index)
    k = 0
    for n in range(N + 1):
        while k < L and pos[k] <= n:
            stmts.append(("leaf", k))
            k += 1
        if n < N:
            stmts.append(("node", n))
    return stmts, names, roots_of


def table_to_julia_str(table: NodeTable, ids=None, name: str = "eval_graph!") -> str:
    stmts, names, roots_of = _emit(table, ids, "julia")
    L = table.n_leaf
    body = []
    for kind, i in stmts:
        v = i if kind == "leaf" else L + i
        if kind == "leaf":
            body.append(f"    {names[v]} = leafVal[{i + 1}]\n")
        else:
            body.append(f"    {names[v]} = {_expr(table, i, names, 'julia')}\n")
        if v in roots_of:
            body.append(f"    root[{roots_of[v] + 1}] = {names[v]}\n")
    return f"\nfunction {name}(root::AbstractVector, leafVal::AbstractVector)\n" + "".join(body) + "end"


def table_to_Cstr(table: NodeTable, ids=None, name: str = "eval_graph", ctype: str = "double ") -> str:
    stmts, names, roots_of = _emit(table, ids, "c")
    L = table.n_leaf
    declare = f"    {ctype}"
    body = []
    for kind, i in stmts:
        v = i if kind == "leaf" else L + i
        declare += f" {names[v]},"
        if kind == "leaf":
            body.append(f"    {names[v]} = leafVal[{i}];\n")
        else:
            body.append(f"    {names[v]} = {_expr(table, i, names, 'c')};\n")
        if v in roots_of:
            body.append(f"    root[{roots_of[v]}] = {names[v]};\n")
    declare = declare[:-1] + ";\n"
    return f"\nvoid {name}({ctype}*root, {ctype}*leafVal)\n{{\n" + declare + "".join(body) + "}"


def table_to_python_str(table: NodeTable, ids=None, name: str = "eval_graph",
                        in_place: bool = False, n_graphs: Optional[int] = None) -> str:
    stmts, names, roots_of = _emit(table, ids, "python")
    L = table.n_leaf
    body = []
    for kind, i in stmts:
        v = i if kind == "leaf" else L + i
        if kind == "leaf":
            body.append(f"    {names[v]} = leafVal[:, {i}]\n")
        else:
            body.append(f"    {names[v]} = {_expr(table, i, names, 'python')}\n")
        if v in roots_of:
            body.append(f"    root[:, {roots_of[v]}] = {names[v]}\n")
    if in_place:
        head = f"def {name}(root, leafVal):\n"
    else:
        ng = table.n_root if n_graphs is None else n_graphs
        head = ("import torch\n" f"def {name}(leafVal):\n"
                f"    root = torch.empty(leafVal.shape[0], {ng}, dtype=leafVal.dtype, device=leafVal.device)\n")
    return head + "".join(body) + "    return root\n\n"


# graph-level wrappers with the reference's signatures ---------------------- #
def to_julia_str(graphs, root=None, name: str = "eval_graph!"):
    """static.jl:98-133: returns ``(text, leafmap)``."""
    table, leafmap, ids = lower(graphs, root)
    return table_to_julia_str(table, ids, name), leafmap


def julia_to_C_typestr(datatype) -> str:
    """static.jl:134-153: the C spelling of the Julia weight type ``to_Cstr`` / ``compile_C`` take as ``datatype``.  Accepts the
    Julia names (``"Float64"``, ``"Float32"``, ``"Int64"``, ``"Int32"``, ``"ComplexF32"``, ``"ComplexF64"``), numpy dtypes /
    Python types of the same meaning, an ``"Array{T}"`` / ``"Vector{T}"`` of one of them (a pointer), or an already spelled C
    type ending in a blank or ``*``.  Anything else: ``error("Unsupported type")`` as in the reference."""
    table = {"Float64": "double ", "Float32": "float ", "Int64": "long long ", "Int32": "int ",
             "ComplexF32": "complex float ", "ComplexF64": "complex double ",
             "float64": "double ", "float32": "float ", "int64": "long long ", "int32": "int ",
             "complex64": "complex float ", "complex128": "complex double ", "float": "double ", "int": "long long ", "complex": "complex double "}
    if isinstance(datatype, str):
        if datatype.endswith((" ", "*")) and datatype.strip("* ") in {v.strip() for v in table.values()}:
            return datatype
        for wrap in ("Array{", "Vector{", "Matrix{"):
            if datatype.startswith(wrap) and datatype.endswith("}"):
                return julia_to_C_typestr(datatype[len(wrap):-1].split(",")[0].strip()) + "*"
        if datatype in table:
            return table[datatype]
        raise ValueError("Unsupported type")
    name = getattr(datatype, "__name__", None) or str(getattr(datatype, "name", datatype))
    if name in table:
        return table[name]
    raise ValueError("Unsupported type")


def to_Cstr(graphs, root=None, datatype="Float64", name: str = "eval_graph"):
    """static.jl:155-197: returns ``(text, leafmap)``; ``datatype`` as ``julia_to_C_typestr`` takes it (default: the
    reference's ``_dtype.weight`` = Float64)."""
    table, leafmap, ids = lower(graphs, root)
    return table_to_Cstr(table, ids, name, julia_to_C_typestr(datatype)), leafmap


def to_python_str(graphs, root=None, name: str = "eval_graph", in_place: bool = False):
    """compiler_python.jl:9-52: returns ``(text, leafmap)``."""
    graphs = list(graphs)
    table, leafmap, ids = lower(graphs, root)
    return table_to_python_str(table, ids, name, in_place, len(graphs)), leafmap
