"""Known answer of the reference's own test "Taylor AD of Sigma FeynmanGraph" (test/taylor.jl:97-113) as a travelling fixture.

The reference asserts, for the 2nd-order self-energy and the eight counter-term orders (2, GOrder, VerOrder) in
{(2,0,0), (2,0,1), (2,0,2), (2,1,0), (2,1,1), (2,2,0), (2,1,2), (2,2,2)}:

    eval!(GV.diagsGV(:sigma, 2, GOrder, VerOrder)[1][k]) == eval!(taylorexpansion!(diagsGV(:sigma, 2, 0, 0))[k].coeffs[[GOrder, VerOrder]])

with all leaves 1, variable x attached to the fermionic and y to the bosonic propagators (orders [2, 2]).  The left side
needs nothing but the catalogs: all leaves 1 gives sum over the diagrams of Sigma2_<VerOrder>_<GOrder>.diag of
SymFactor * sum(SpinFactor) per pair of external times (GV.jl:60 names the file Sigma$(order)_$(VerOrder)_$(GOrder); the
FeynmanGraph reader, readfile.jl:636-714, builds linear_combination(feynman_diagram(..., factor=symfactor, is_signed=true),
spinfactors)).  The right side is our restated Taylor pass (producers/taylor.py) on our restated reader's graph of
Sigma2_0_0.diag.  This script computes the left side from the catalog text, builds the right side, checks them equal, and
writes (a) the 16 expected numbers and (b) the node table whose 16 roots are those Taylor coefficients, so that the CPU
oracle and the device back ends can be held to the reference's own numbers without the reference present.

Needs /root/reference (this container only).  Run: python tests/golden/make_gv_counterterm_kat.py
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
RD = "/root/reference/src/frontend/GV_diagrams"
ORDERS = [(2, 0, 0), (2, 0, 1), (2, 0, 2), (2, 1, 0), (2, 1, 1), (2, 2, 0), (2, 1, 2), (2, 2, 2)]     # test/taylor.jl:98

import oracle  # noqa: E402
from feynmandiagram_jl_amd.graph import PostOrderDFS, isleaf  # noqa: E402
from feynmandiagram_jl_amd.lowering import lower  # noqa: E402
from feynmandiagram_jl_amd.producers import gv, taylor  # noqa: E402


def catalog_sums(path):
    """{(tau of the incoming external leg, tau of the outgoing one): sum over diagrams of SymFactor * sum(SpinFactor)}"""
    txt = open(path).read().split("\n\n")
    ext = [int(x) for x in re.findall(r"[-+]?\d+", [l for l in txt[0].split("\n") if "ExtTauIndex" in l][0])]
    out = {}
    for blk in txt[1:]:
        lines = [l for l in blk.split("\n") if l.strip()]
        if not lines or "Permutation" not in lines[0]:
            continue
        perm = [int(x) for x in lines[1].split()]
        sym = float(lines[3])
        tau = [int(x) for x in lines[7].split()]
        spin = [int(x) for x in lines[-1].split()]
        key = (tau[ext[0]], tau[perm.index(ext[0])])
        out[key] = out.get(key, 0.0) + sym * sum(spin)
    return out


def build():
    graphs = gv.diagsGV("sigma", 2, RD)
    taylor.set_variables("x y", orders=[2, 2])
    dep = {}
    for g in graphs:
        for n in PostOrderDFS(g):
            if isleaf(n):       # propagator_var = ([true, false], [false, true]): x on fermionic, y on bosonic lines (test/taylor.jl:106)
                dep[n.id] = [isinstance(n.properties, gv.BareGreenId), isinstance(n.properties, gv.BareInteractionId)]
    series = []
    for g in graphs:
        t = taylor.taylorexpansion(g, dep)
        series.append(t[0] if isinstance(t, tuple) else t)
    roots, expected, labels = [], [], []
    for o in ORDERS:
        want = catalog_sums(f"{RD}/groups_sigma/Sigma{o[0]}_{o[2]}_{o[1]}.diag")
        for g, s in zip(graphs, series):
            key = tuple(x - 1 for x in g.properties.extT)
            roots.append(s.coeffs[(o[1], o[2])])
            expected.append(want[key])
            labels.append(f"order {o} extT {g.properties.extT}")
    table, _, _ = lower(roots, name="gv_sigma2_taylor_x2_y2_coefficients")
    return table.normalized(), expected, labels


def main():
    table, expected, labels = build()
    got = oracle.eval_static(table, np.ones((1, table.n_leaf)))[0].tolist()
    assert got == expected, list(zip(labels, got, expected))
    table.save(os.path.join(HERE, "gv_sigma2_counterterm_kat.npz"))
    json.dump({"source": "test/taylor.jl:97-113; catalogs src/frontend/GV_diagrams/groups_sigma/Sigma2_<VerOrder>_<GOrder>.diag",
               "orders": [list(o) for o in ORDERS], "labels": labels, "expected": expected},
              open(os.path.join(HERE, "gv_sigma2_counterterm_kat.json"), "w"), indent=1)
    print(table.stats())
    for l, e in zip(labels, expected):
        print(l, e)


if __name__ == "__main__":
    main()
