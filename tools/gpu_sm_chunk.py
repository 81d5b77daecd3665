"""GPU dev tool: sample-major (row-major [B, L]) input through the ISA kernel for several transposition chunk sizes."""
def _need_dev_build():
    """This tool steers the library through FDG_* environment variables AFTER it is loaded: only the dev build (make -C feynmandiagram.jl_amd/csrc dev;
    FDG_LIBRARY=.../libfdg_dev.so) reads them then -- the product build snapshots the supported ones once per process and would compare a configuration
    with itself.  Fail loudly instead."""
    import os, sys
    if not os.environ.get("FDG_LIBRARY", "").endswith("libfdg_dev.so"):
        sys.exit(os.path.basename(__file__) + ": needs the dev build (make -C feynmandiagram.jl_amd/csrc dev; export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so): "
                 "the product library reads FDG_* once per process, so the switches this tool flips would be silent no-ops")


_need_dev_build()

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
for name, B in (("gv_sigma4_taylor2", 4_000_000), ("gv_sigma5", 2_000_000), ("gv_sigma4", 8_000_000)):
    t = workloads.get(name)
    f = fd.compile_table(t, specialize="isa")
    L, R = t.n_leaf, t.n_root
    leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
    capi.fill_uniform_device(leaf.data_ptr(), B, L, L, 1, 11, 0, torch.cuda.current_stream().cuda_stream)
    root = torch.empty((B, R), dtype=torch.float64, device=dev)
    for mb in (512, 256, 128, 64, 32, 16, 8):
        os.environ["FDG_SM_CHUNK_MB"] = str(mb)
        for _ in range(3): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(f"{name} sample_major chunk {mb} MB: {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s", flush=True)
    del leaf, root
