"""Copies the node tables the product ships (workloads.get("gv_*"), workloads.leafstates) from the golden fixtures
into feynmandiagram.jl_amd/data/, without the fixtures' input/expected-output vectors: the package must not depend
on a test directory.  Run after make_gv_tables.py: python tests/golden/make_package_data.py
(tests/test_host_api.py::test_package_tables_equal_golden_fixtures keeps the two in step)."""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(os.path.dirname(os.path.dirname(HERE)), "feynmandiagram.jl_amd", "data")
TABLE_KEYS = ("n_leaf", "op", "power", "child_off", "child_idx", "child_fac", "root_slot", "name", "leaf_pos",
              "leaf_base", "leaf_dorder", "sched_group")
NAMES = ("gv_sigma4", "gv_sigma5", "gv_sigma6", "gv_sigma4_taylor2", "gv_sigma5_taylor2")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for n in NAMES:
        z = np.load(os.path.join(HERE, n + ".npz"))
        np.savez_compressed(os.path.join(DST, n + ".npz"), **{k: z[k] for k in TABLE_KEYS if k in z.files})
    for n in ("gv_sigma4", "gv_sigma5"):
        z = np.load(os.path.join(HERE, n + "_leafstates.npz"))
        np.savez_compressed(os.path.join(DST, n + "_leafstates.npz"), **{k: z[k] for k in z.files})
    print(sorted(os.listdir(DST)))
