"""Generates the committed fixtures under tests/golden/.

Two kinds of data, kept apart:

* ``kat.json`` -- known answers copied from the reference's own tests (values
  only; see the ``source`` field of each entry for file:line).
* ``*.npz`` -- node tables plus seeded leaf inputs and the roots our CPU
  restatement (oracle/fdg_oracle.c) produces for them.  SELF-GENERATED, NOT
  PRODUCED BY JULIA: Julia is not installed in this environment, so the
  reference evaluator itself cannot be run.  They guard
  against regressions of the oracle and give the GPU tests fixed vectors.

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from feynmandiagram_jl_amd import workloads  # noqa: E402


def main():
    kat = [
        dict(name="compiler_jl", source="test/compiler.jl:4-15,19-28", leaf=[1.0, 2.0], expect=[4.5],
             note="eval_graph!(root, leaf) ≈ (leaf[1]+leaf[2])*1.5 and returns that value"),
        dict(name="evaluation_g3_g4_g5", source="test/computational_graph.jl:874-887", leaf="ones",
             expect=[26.0, 27.0, 702.0], note="exact =="),
        dict(name="taylor_getdiagram_spin0.5", source="test/taylor.jl:115-161,181-202", leaf="ones",
             expect=[(-2 + 0.5) / (2 * math.pi) ** 3], note="≈ (rtol sqrt(eps))"),
        dict(name="front_end_getdiagram_spin1.0", source="test/front_end.jl:287-309", leaf="ones",
             expect=[(-2 + 1.0) / (2 * math.pi) ** 3], note="≈"),
        dict(name="sigma2_all_ones", source="assets/sigma_o2.svg + README.md:59-72 (structure); value by restated evaluation",
             leaf="ones", expect=[1.0, -1.0], note="not a Julia-produced value"),
    ]
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, ensure_ascii=False)

    kat.append(dict(name="gv_sigma_all_ones", source="src/frontend/GV_diagrams/groups_sigma/Sigma{4,5,6}_0_0.diag: sum over diagrams of SymFactor*sum(SpinFactor), per external-tau group (tests/golden/make_gv_tables.py)",
                    leaf="ones", expect={"4": [21.0, 3.0], "5": [-31.0, -77.0], "6": [233.0, 167.0]},
                    note="computed from the catalog text, independent of the reader/optimizer restatements"))
    kat.append(dict(name="parquet_sigma_diagram_counts", source="test/front_end.jl:600-652 with src/frontend/parquet/benchmark/diagram_count.jl:53-66 "
                    "(spin 2, bosonic signs, filter [NoHartree, Girreducible]): all leaves 1 => (-1)^n * count_sigma_G2v(n, 2)",
                    leaf="ones", expect={"1": -1.0, "2": 3.0, "3": -18.0, "4": 171.0}, note="exact (integers); tests/test_parquet.py"))
    kat.append(dict(name="parquet_vertex3_and_polarization_diagram_counts", source="test/front_end.jl:701-755 (count_ver3_G2v), :758-826 "
                    "(count_polar_G2v, count_polar_g2v_noFock, count_polar_g2v_noFock_upup); src/frontend/parquet/benchmark/diagram_count.jl",
                    leaf="ones", expect={"ver3_G2v": [1, 10, 109], "polar_G2v": [2, 2, 20, 218], "polar_g2v_noFock": [2, 2, 32, 326],
                                         "polar_g2v_noFock_upup": [2, 2, 28, 274]},
                    note="num * (-1)^n resp. num * spin * (-1)^(n-1); exact; tests/test_parquet.py"))
    kat.append(dict(name="parquet_against_gv_catalog_sums", source="src/frontend/GV_diagrams/groups_sigma/Sigma{2..6}_0_0.diag, groups_vertex4/Vertex4{1..4}_0_0.diag, "
                    "groups_charge|groups_spin/Polar{1..5}_0_0.diag: sums of SymFactor*SpinFactor computed from the catalog text",
                    leaf="ones", expect={"sigma_dynamic_instant": {"2": [1, -1], "3": [-5, 1], "4": [21, 3], "5": [-77, -31], "6": [233, 167]},
                                         "vertex4_updown": [2, -9, 40, -168], "polar_charge": [-2, 6, -10, -42, 558], "polar_spin": [-2, 6, -18, 46, -66]},
                    note="the Parquet builder's fermionic graphs give the same sums (self-energy: times -1); tests/test_parquet.py"))
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, ensure_ascii=False)
    for name, B, seed in (("sigma2", 257, 1234), ("synthetic_small", 64, 1234), ("sigma4_standin", 16, 1234),
                          ("sigma4_worstcase", 8, 1234), ("gv_sigma5", 16, 1234), ("gv_sigma4_taylor2", 32, 1234)):
        t = workloads.get(name)
        leaf = oracle.philox_uniform(B, t.n_leaf, seed)
        root = oracle.eval_static(t, leaf)
        root_interp = oracle.eval_interp(t, leaf)
        assert np.array_equal(root, oracle.eval_static_numpy(t, leaf))
        tn = t.normalized()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), n_leaf=np.int64(tn.n_leaf), op=tn.op, power=tn.power,
                            child_off=tn.child_off, child_idx=tn.child_idx, child_fac=tn.child_fac,
                            root_slot=tn.root_slot, name=np.array(tn.name),
                            leaf_pos=tn.leaf_positions().astype(np.uint32),
                            seed=np.int64(seed), leaf=leaf, root_static=root, root_interp=root_interp,
                            **({k: np.load(os.path.join(HERE, f"{name}.npz"))[k] for k in ("leaf_base", "leaf_dorder", "sched_group")}
                               if name.endswith("taylor2") else {}))
        print(name, t.stats(), root[0])
    # the 4-loop Parquet self-energy (configs 3 and 4) from the restated front end: tables built on the fly, the
    # vectors pin the builder + optimize! + lowering + oracle chain against regressions
    for name, B, seed in (("parquet_sigma4", 64, 1234), ("parquet_sigma4_taylor2", 32, 1234)):
        t = workloads.get(name)
        leaf = oracle.philox_uniform(B, t.n_leaf, seed)
        root = oracle.eval_static(t, leaf)
        assert np.array_equal(root, oracle.eval_static_numpy(t, leaf))
        tn = t.normalized()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), n_leaf=np.int64(tn.n_leaf), op=tn.op, power=tn.power,
                            child_off=tn.child_off, child_idx=tn.child_idx, child_fac=tn.child_fac, root_slot=tn.root_slot,
                            name=np.array(tn.name), leaf_pos=tn.leaf_positions().astype(np.uint32), seed=np.int64(seed), leaf=leaf,
                            root_static=root, root_interp=oracle.eval_interp(t, leaf),
                            **({} if tn.sched_group is None else {"sched_group": tn.sched_group}))
        print(name, t.stats(), root[0])


if __name__ == "__main__":
    main()
