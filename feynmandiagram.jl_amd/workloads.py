"""Named graphs used by bench.py, the tests and ``__graft_entry__``.

Each maps to a config of BASELINE.json:
  sigma2            configs 1-2: optimized 2-loop Parquet self-energy (fixture)
  sigma4_standin    config 3: seeded stand-in for the 4-loop Parquet self-energy
                    (N ~ 10^4 nodes, L = 300, parquet-recursion sharing; the real
                    graph needs the Julia front end)
  sigma4_worstcase  same size, operands drawn uniformly at random (no locality)
  gv_sigma4_taylor2, gv_sigma5_taylor2
                    config 4 on real reference data: the GV 4-/5-loop self-energy with Taylor-mode
                    AD counterterms of order 2 in the coupling (restated taylorAD + optimize!;
                    roots = orders 0,1,2 x {instant, dynamic}); N = 7 373 / 115 588 nodes
  gv_sigma4/5/6     real reference data: GV self-energy catalogs of order 4/5/6
                    read from the reference's .diag files and run through the
                    restated optimize! (tests/golden/make_gv_tables.py; shipped in data/)
  sigma4_taylor_standin  config 4: the same enlarged x3 with 2 % Power{2} nodes
  parquet_sigma{2,3,4}[_dyn|_insdyn][_taylor2]
                    configs 1-4 from the restated Parquet front end (producers/parquet.py): ``Parquet.build(DiagPara(type=SigmaDiag,
                    innerLoopNum=n, hasTau=true, filter=[NoHartree]))`` -> ``optimize!``; interaction ChargeCharge
                    Instant (the reference's default; 4 loops: L 84, N 1 325, R 4), ``_dyn`` Dynamic (L 175, N 4 819,
                    R 7), ``_insdyn`` both (L 312, N 20 147, R 8: the size BASELINE.json quotes as "~10^4 nodes" lies
                    between the last two); ``_taylor2``: taylorAD of order 2 in the coupling + optimize! (config 4).
                    Built on the fly, no data file.  ``parquet_sigma5`` (L 274, N 11 407, R 5) and above use the fully
                    irreducible vertex of the GV catalogs (data/vertex4I<n>.npz).
  parquet_ver4_4    the graph example/benchmark.jl builds: ``Parquet.vertex4(DiagPara(type=Ver4Diag, innerLoopNum=4))`` +
                    optimize! (L 984, N 44 854, 180 roots)
  gv_ver4_4         the graph example/benchmark_GV.jl builds: ``GV.diagsGV_ver4(4)`` + optimize! (catalog Vertex44_0_0.diag,
                    1 190 Hugenholtz diagrams: L 1 514, N 31 803, 26 roots; tests/golden/make_vertex4_catalogs.py)
  synthetic_small   a 1000-node graph for quick parity runs
"""
from __future__ import annotations

import functools
import os

import numpy as np

from .nodetable import NodeTable, OP_POWER, OP_PROD, from_program, synthetic_parquet_like

# node tables derived from the reference's GV catalogs (tests/golden/make_gv_tables.py, make_package_data.py)
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

PREBUILT = ("sigma2", "synthetic_small", "sigma4_standin", "sigma4_worstcase", "sigma4_taylor_standin", "gv_sigma4",
            "gv_sigma5", "gv_sigma6", "gv_sigma4_taylor2", "gv_sigma5_taylor2", "parquet_sigma2", "parquet_sigma3",
            "parquet_sigma4", "parquet_sigma4_dyn", "parquet_sigma4_insdyn", "parquet_sigma4_taylor2",
            "parquet_sigma4_dyn_taylor2", "parquet_sigma4_insdyn_taylor2", "parquet_sigma5", "parquet_ver4_4", "gv_ver4_4")
PREBUILT_HIP = ("sigma2", "synthetic_small", "sigma4_standin", "sigma4_worstcase", "gv_sigma5", "gv_sigma4_taylor2")


@functools.lru_cache(maxsize=None)
def get(name: str) -> NodeTable:
    if name == "sigma2":
        from .fixtures import sigma2_graphs
        from .lowering import lower
        t, _, _ = lower(sigma2_graphs()[0], name="sigma2")
        t.name = "sigma2_parquet_nohartree_optimized"
        return t
    if name == "sigma4_standin":
        return synthetic_parquet_like(10000, 300, 2, seed=20241220)
    if name == "sigma4_worstcase":
        return synthetic_parquet_like(10000, 300, 2, seed=20241220, structure="random")
    if name == "synthetic_small":
        return synthetic_parquet_like(1000, 64, 2, seed=7, structure="random")
    if name.startswith("gv_sigma") or name.startswith("gv_ver4"):
        return NodeTable.load(os.path.join(DATA, name + ".npz"))
    if name == "sigma4_taylor_standin":
        return _with_powers(synthetic_parquet_like(30000, 300, 6, seed=20241221), 0.02, 99)
    if name.startswith("parquet_"):
        return _parquet(name)
    raise KeyError(name)


def parquet_graphs(name: str):
    """The optimized graphs of a ``parquet_sigma<n>[_dyn|_insdyn]`` / ``parquet_ver4_<n>`` workload and the front end's table rows."""
    import re

    from .producers import optimize, parquet as pq
    m = re.fullmatch(r"parquet_(sigma|ver4_)(\d)(_dyn|_insdyn)?", name)
    if not m:
        raise KeyError(name)
    types = {None: (pq.Instant,), "_dyn": (pq.Dynamic,), "_insdyn": (pq.Instant, pq.Dynamic)}[m.group(3)]
    para = pq.DiagPara(type=pq.SigmaDiag if m.group(1) == "sigma" else pq.Ver4Diag, innerLoopNum=int(m.group(2)), hasTau=True,
                       filter=(pq.NoHartree,), interaction=(pq.Interaction(pq.ChargeCharge, types),))
    rows = pq.build(para)
    graphs = [r["diagram"] for r in rows]
    optimize.optimize_(graphs)
    return graphs, rows


@functools.lru_cache(maxsize=None)
def _parquet_lowered(name: str):
    """(table, leafmap) of a parquet workload."""
    from .producers import gv, optimize, taylor
    from .lowering import lower
    taylor2 = name.endswith("_taylor2")
    graphs, _ = parquet_graphs(name[:-len("_taylor2")] if taylor2 else name)
    if not taylor2:
        t, leafmap, _ = lower(graphs, name=name)
        return t.normalized(), leafmap
    groups = {}
    d = taylor.taylorAD(graphs, [2], [lambda pr: isinstance(pr, gv.BareInteractionId)], groups=groups)
    allg = [g for o in sorted(d) for g in d[o]]              # roots: orders 0, 1, 2 x the rows of the self-energy
    optimize.optimize_(allg)
    t, leafmap, _ = lower(allg, name=name, groups=groups)
    return t.normalized(), leafmap


def _parquet(name: str) -> NodeTable:
    return _parquet_lowered(name)[0]


def _parquet_leafstates(name: str):
    """``FrontEnds.leafstates(leaf_maps, maxloopNum)`` (frontends.jl:178-232) of a parquet workload whose interaction is
    instantaneous (the leaf formulas of example/benchmark.jl:58-127 cover the fermionic propagator and the
    instantaneous Yukawa interaction with its counter-terms)."""
    if "_dyn" in name or "_insdyn" in name:
        return None
    from . import frontends
    t, leafmap = _parquet_lowered(name)
    n_loop = max(len(leafmap[i + 1].properties.extK) for i in range(t.n_leaf))
    (val, typ, orders, tin, tout, loopidx), basis = frontends.leafstates([leafmap], n_loop)
    lorder = [int(orders[0][i][0]) if (typ[0][i] == 2 and len(orders[0][i]) == 1) else 0 for i in range(t.n_leaf)]
    return dict(leaf_type=np.array(typ[0], np.int32), leaf_order=np.array(lorder, np.int32), tau_in=np.array(tin[0], np.int32),
                tau_out=np.array(tout[0], np.int32), loop_index=np.array(loopidx[0], np.int32), basis=np.array(basis, np.float64),
                n_tau=np.int64(max(max(tin[0]), max(tout[0]))))


PREBUILT_MC = ("gv_sigma4", "gv_sigma4_taylor2", "gv_sigma5", "parquet_sigma4", "parquet_sigma4_taylor2")     # one-kernel Monte-Carlo step assembled by build()


def leafstates(name: str):
    """The ``FrontEnds.leafstates`` tables of a GV workload, in leafVal order (``leaf_type``, ``leaf_order``, ``tau_in``,
    ``tau_out``, ``loop_index``, ``basis``, ``n_tau``), or None.  For the Taylor-expanded graphs a leaf is (leaf of the
    original graph, derivative order in the coupling): the fixture of the original graph re-indexed."""
    if name.startswith("parquet_"):
        return _parquet_leafstates(name)
    base = name[:-len("_taylor2")] if name.endswith("_taylor2") else name
    path = os.path.join(DATA, base + "_leafstates.npz")
    if not os.path.exists(path):
        return None
    z = dict(np.load(path))
    if name.endswith("_taylor2"):
        zt = np.load(os.path.join(DATA, name + ".npz"))
        for k in ("leaf_type", "tau_in", "tau_out", "loop_index"):
            z[k] = z[k][zt["leaf_base"]]
        z["leaf_order"] = np.where(z["leaf_type"] == 2, zt["leaf_dorder"], 0).astype(np.int32)
    return z


def _with_powers(t: NodeTable, frac: float, seed: int) -> NodeTable:
    """Turn a fraction of two-child Prod nodes with equal children ... into
    Power{2}: here simply re-type random single-use Prod nodes as Power{2} of
    their first child (keeps the table topologically valid)."""
    rng = np.random.default_rng(seed)
    nodes = []
    for n in range(t.n_node):
        ch = t.children(n)
        if int(t.op[n]) == OP_PROD and rng.random() < frac * 1.5:
            nodes.append((OP_POWER, 2, [ch[0]]))
        else:
            nodes.append((int(t.op[n]), int(t.power[n]), ch))
    return from_program(t.n_leaf, nodes, [int(r) for r in t.root_slot], t.name + "_pow")
