"""Times fdg_accumulate_device on the optimizing back end with and without the fused
in-register accumulation (FDG_ISA_NO_FUSED_ACC=1).  Run on the GPU box."""
def _need_dev_build():
    """This tool steers the library through FDG_* environment variables AFTER it is loaded: only the dev build (make -C feynmandiagram.jl_amd/csrc dev;
    FDG_LIBRARY=.../libfdg_dev.so) reads them then -- the product build snapshots the supported ones once per process and would compare a configuration
    with itself.  Fail loudly instead."""
    import os, sys
    if not os.environ.get("FDG_LIBRARY", "").endswith("libfdg_dev.so"):
        sys.exit(os.path.basename(__file__) + ": needs the dev build (make -C feynmandiagram.jl_amd/csrc dev; export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so): "
                 "the product library reads FDG_* once per process, so the switches this tool flips would be silent no-ops")


_need_dev_build()

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads

dev = torch.device("cuda:0")
for name, B in (("sigma2", 1 << 26), ("gv_sigma4", 1 << 22), ("gv_sigma4_taylor2", 1 << 22), ("gv_sigma5", 1 << 20), ("gv_sigma6", 1 << 18)):
    t = workloads.get(name)
    leaf = torch.rand((t.n_leaf, B), dtype=torch.float64, device=dev).t()
    w = torch.rand(B, dtype=torch.float64, device=dev)
    root = torch.empty((t.n_root, B), dtype=torch.float64, device=dev).t()
    fs = {}
    for fused in (True, False):
        if fused:
            os.environ.pop("FDG_ISA_NO_FUSED_ACC", None)
        else:
            os.environ["FDG_ISA_NO_FUSED_ACC"] = "1"
        fs[fused] = fd.compile_table(t, specialize="isa")
    best = {}
    for rnd in range(6):
        for fused in (True, False):
            f = fs[fused]
            if fused:
                os.environ.pop("FDG_ISA_NO_FUSED_ACC", None)
            else:
                os.environ["FDG_ISA_NO_FUSED_ACC"] = "1"
            for what in ("eval", "acc"):
                fn = (lambda: f(root, leaf)) if what == "eval" else (lambda: f.accumulate(leaf, w))
                fn(); fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
                k = (fused, what)
                best[k] = min(best.get(k, 1e9), dt)
    for (fused, what), dt in best.items():
        print(f"{name:20s} fused={int(fused)} {what:4s} {dt*1e3:8.3f} ms  {B/dt:.3e} samples/s", flush=True)
