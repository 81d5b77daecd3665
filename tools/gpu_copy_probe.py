"""GPU dev tool: rate of the library's copy kernel (bench.py's measured_copy) with and without non-temporal accesses."""
def _need_dev_build():
    """This tool steers the library through FDG_* environment variables AFTER it is loaded: only the dev build (make -C feynmandiagram.jl_amd/csrc dev;
    FDG_LIBRARY=.../libfdg_dev.so) reads them then -- the product build snapshots the supported ones once per process and would compare a configuration
    with itself.  Fail loudly instead."""
    import os, sys
    if not os.environ.get("FDG_LIBRARY", "").endswith("libfdg_dev.so"):
        sys.exit(os.path.basename(__file__) + ": needs the dev build (make -C feynmandiagram.jl_amd/csrc dev; export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so): "
                 "the product library reads FDG_* once per process, so the switches this tool flips would be silent no-ops")


_need_dev_build()

import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
for rep in range(2):
    for nt in (0, 1):
        if nt: os.environ.pop("FDG_COPY_PLAIN", None)
        else: os.environ["FDG_COPY_PLAIN"] = "1"
        print("nt" if nt else "plain", "%.0f GB/s" % bench.measured_copy(dev), flush=True)
