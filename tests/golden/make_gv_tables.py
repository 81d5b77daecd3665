"""Derives node tables from the reference's GV ``.diag`` catalogs (data files of
the reference: src/frontend/GV_diagrams/groups_sigma/Sigma{4,5,6}_0_0.diag).

    diagsGV(:sigma, order)  ->  optimize!  ->  Compilers lowering  ->  .npz

All three steps are our restatements (feynmandiagram.jl_amd/producers/gv.py, optimize.py,
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
from feynmandiagram_jl_amd.producers import gv, optimize  # noqa: E402
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


def taylor_tables():
    """BASELINE.json config 4: self-energy graphs with Taylor-mode AD counterterms of order 2 in the
    coupling (every BareInteractionId leaf depends on the expansion variable, README.md:83).  Checked
    independently of the Taylor restatement: the original graph evaluated at V = V0 + x V1 + x^2 V2
    must equal c0 + x c1 + x^2 c2 up to O(x^3)."""
    from feynmandiagram_jl_amd.producers import taylor
    for order in (4, 5):
        graphs = gv.diagsGV("sigma", order, RD)
        optimize.optimize_(graphs)
        t0, lm0, _ = lower(graphs)
        groups = {}
        d = taylor.taylorAD(graphs, [2], [lambda pr: isinstance(pr, gv.BareInteractionId)], groups=groups)
        allg = [g for o in sorted(d) for g in d[o]]          # roots: (c0_ins, c0_dyn, c1_ins, c1_dyn, c2_ins, c2_dyn)
        optimize.optimize_(allg)
        t, lm, _ = lower(allg, name=f"gv_sigma{order}_taylor2_optimized", groups=groups)
        # leaf bookkeeping: which original leaf and which derivative order each leaf of the enlarged graph is
        key0 = {}
        for i in range(t0.n_leaf):
            key0[lm0[i + 1].properties.equiv_key()] = i
        base = np.array([key0[lm[i + 1].properties.equiv_key()] for i in range(t.n_leaf)], dtype=np.int32)
        dord = np.array([int(lm[i + 1].orders[0]) if len(lm[i + 1].orders) == 1 else 0 for i in range(t.n_leaf)], dtype=np.int32)
        rng = np.random.default_rng(order)
        v = [rng.uniform(0.5, 1.5, size=(3, t0.n_leaf)) for _ in range(1)][0]       # V0, V1, V2 per original leaf
        x = 1e-3
        is_v = np.array([isinstance(lm0[i + 1].properties, gv.BareInteractionId) for i in range(t0.n_leaf)])
        leaf_x = v[0] + np.where(is_v, x * v[1] + x * x * v[2], 0.0)
        f_x = oracle.eval_static(t0, leaf_x[None, :])[0]
        big = np.array([[v[dord[i], base[i]] for i in range(t.n_leaf)]])
        c = oracle.eval_static(t, big)[0].reshape(3, 2)
        series = c[0] + x * c[1] + x * x * c[2]
        scale = np.abs(c[0]) + 1.0
        assert np.all(np.abs(series - f_x) <= 50 * x ** 3 * scale * 1e3), (order, series, f_x)
        assert np.all(np.abs((c[0] + x * c[1]) - f_x) > np.abs(series - f_x)), "second order must improve on first"
        tn = t.normalized()
        np.savez_compressed(os.path.join(HERE, f"gv_sigma{order}_taylor2.npz"), n_leaf=np.int64(tn.n_leaf), op=tn.op,
                            power=tn.power, child_off=tn.child_off, child_idx=tn.child_idx, child_fac=tn.child_fac,
                            root_slot=tn.root_slot, name=np.array(tn.name), leaf_pos=tn.leaf_positions().astype(np.uint32),
                            sched_group=tn.sched_group, leaf_base=base, leaf_dorder=dord)
        print("taylor2", order, t.stats(), "series err", np.abs(series - f_x), "first-order err", np.abs((c[0] + x * c[1]) - f_x))


def leafstates_fixture(order=4):
    """FrontEnds.leafstates(leaf_maps, maxloopNum) (frontends.jl:178-232) of the optimized GV self-energy of that
    order, in leafVal index order: what the device leaf kernels and the one-kernel Monte-Carlo step consume."""
    from feynmandiagram_jl_amd import FrontEnds
    graphs = gv.diagsGV("sigma", order, RD)
    optimize.optimize_(graphs)
    t, leafmap, _ = lower(graphs)
    w = np.load(os.path.join(HERE, f"gv_sigma{order}.npz"))
    assert np.array_equal(t.normalized().child_idx, w["child_idx"])
    (val, typ, orders, tin, tout, loopidx), basis = FrontEnds.leafstates([leafmap], order + 1)
    lorder = [orders[0][i][0] if typ[0][i] == 1 else orders[0][i][1] for i in range(t.n_leaf)]
    n_tau = max(max(tin[0]), max(tout[0]))
    np.savez_compressed(os.path.join(HERE, f"gv_sigma{order}_leafstates.npz"), leaf_type=np.array(typ[0], np.int32),
                        leaf_order=np.array(lorder, np.int32), tau_in=np.array(tin[0], np.int32), tau_out=np.array(tout[0], np.int32),
                        loop_index=np.array(loopidx[0], np.int32), basis=np.array(basis, np.float64), n_tau=np.int64(n_tau))
    print(f"leafstates sigma{order}: L", t.n_leaf, "types", np.bincount(typ[0]).tolist(), "n_basis", len(basis), "n_loop", len(basis[0]), "n_tau", n_tau)


if __name__ == "__main__":
    main()
    taylor_tables()
    leafstates_fixture(4)
    leafstates_fixture(5)
