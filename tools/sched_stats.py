"""Host-only: op counts of the allocated ISA program of a workload for a register configuration.
python tools/sched_stats.py [workload ...]   (env FDG_* knobs of fdg_opt.cpp apply)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads

KINDS = {0: "ld_leaf", 1: "ld_lds", 2: "ld_mem", 3: "st_lds", 4: "st_mem", 5: "mul", 6: "add", 7: "mulc", 8: "root", 9: "mov", 10: "ld_acc", 11: "st_acc", 28: "ld_land"}
CFG = {"A": dict(n_reg=120, n_lds=40, n_acc=0), "B": dict(n_reg=120, n_lds=80, n_acc=124)}

def stats(name, cfg="B", **kw):
    t = workloads.get(name)
    g = capi.GraphHandle(t)
    if t.sched_group is not None:
        g.set_schedule_groups(t.sched_group)
    p = dict(CFG[cfg]); p.update(kw)
    t0 = time.time()
    ops, nr, nl, nm = g.opt_program(**p)
    dt = time.time() - t0
    c = np.bincount(ops["kind"], minlength=29)
    d = {KINDS[k]: int(c[k]) for k in KINDS}
    d["ld_leaf"] += d["ld_land"]
    valu = d["mul"] + d["add"] + d["mulc"] + d["mov"]
    return dict(name=name, L=t.n_leaf, valu=valu, ld_leaf=d["ld_leaf"], ld_mem=d["ld_mem"], st_mem=d["st_mem"], ld_lds=d["ld_lds"],
                st_lds=d["st_lds"], ld_acc=d["ld_acc"], st_acc=d["st_acc"], n_mem=nm, sec=round(dt, 2))

if __name__ == "__main__":
    names = sys.argv[1:] or ["sigma4_standin", "gv_sigma5", "gv_sigma6", "gv_sigma5_taylor2", "gv_sigma4_taylor2", "sigma4_worstcase"]
    for n in names:
        for cfg in ("A", "B"):
            print(cfg, stats(n, cfg), flush=True)
