"""Derives node tables from the reference's GV ``.diag`` catalogs (data files of
the reference: src/frontend/GV_diagrams/groups_sigma/Sigma{4,5,6}_0_0.diag).

    diagsGV(:sigma, order)  ->  optimize!  ->  Compilers lowering  ->  .npz

All three steps are our restatements (feynmandiagram.jl_amd/gv.py, optimize.py,
lowering.py); Julia is unavailable, so the tables are NOT checked against the
reference's own graph objects.  What is checked here, independently of those
restatements, is the all-leaves-one value of every root: it must equal the sum
over the catalog's diagrams of SymFactor * sum(SpinFactor), which this script
computes straight from the text file.

Needs /root/reference (this container only); the .npz outputs are committed so
the GPU box never reads the reference.  Run: python tests/golden/make_gv_tables.py
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
RD = "/root/reference/src/frontend/GV_diagrams"

import oracle  # noqa: E402
from feynmandiagram_jl_amd import gv, optimize  # noqa: E402
from feynmandiagram_jl_amd.lowering import lower  # noqa: E402


def direct_all_ones(path):
    """sum over diagrams, grouped by external tau pair, of SymFactor * sum(SpinFactor)."""
    txt = open(path).read().split("\n\n")
    hdr = txt[0]
    ext = [int(x) for x in re.findall(r"[-+]?\d+", [l for l in hdr.split("\n") if "ExtTauIndex" in l][0])]
    out, order = {}, []
    for blk in txt[1:]:
        lines = [l for l in blk.split("\n") if l.strip()]
        if not lines or "Permutation" not in lines[0]:
            continue
        perm = [int(x) for x in lines[1].split()]
        sym = float(lines[3])
        tau = [int(x) for x in lines[7].split()]
        spin = [int(x) for x in lines[-1].split()]
        e0 = ext[0]
        e1 = perm.index(e0)
        key = (tau[e0], tau[e1])
        if key not in out:
            out[key] = 0.0
            order.append(key)
        out[key] += sym * sum(spin)
    return [out[k] for k in order]


def main():
    for order in (4, 5, 6):
        path = f"{RD}/groups_sigma/Sigma{order}_0_0.diag"
        graphs = gv.diagsGV("sigma", order, RD)
        raw, _, _ = lower(graphs)
        v_raw = oracle.eval_static(raw, np.ones((1, raw.n_leaf)))[0]
        optimize.optimize_(graphs)
        t, _, _ = lower(graphs, name=f"gv_sigma{order}_optimized")
        v_opt = oracle.eval_static(t, np.ones((1, t.n_leaf)))[0]
        want = direct_all_ones(path)
        assert list(v_raw) == want == list(v_opt), (order, v_raw, v_opt, want)
        t.save(os.path.join(HERE, f"gv_sigma{order}.npz"))
        print(order, t.stats(), "all-ones roots", v_opt, "== catalog sum", want)


if __name__ == "__main__":
    main()
