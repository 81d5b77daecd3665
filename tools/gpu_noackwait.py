"""GPU dev tool (timing experiment): what would the headline run at if a wave never had to wait for the
acknowledgements of its root stores?  FDG_ISA_DEBUG=noackwait loosens every vmcnt wait by R operations (results may be garbage:
a load may not have landed), so the tile's stores are never waited for.  A/B against the normal kernel on the same batch.
usage: gpu_noackwait.py [workload] [B]"""
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
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32_000_000
t = workloads.get(name); L, R = t.n_leaf, t.n_root
T = (B + 63) // 64
st = torch.cuda.current_stream().cuda_stream
leaf_t = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
root_t = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
capi.fill_uniform_device_tiled(leaf_t.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
leaf_c = torch.empty((L, B), dtype=torch.float64, device=dev)
root_c = torch.zeros((R, B), dtype=torch.float64, device=dev)
capi.fill_uniform_device(leaf_c.data_ptr(), B, L, 1, B, 1234, 0, st)


def timed(fn, n=20, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for dbg in ("", "noackwait", "", "noackwait"):
    if dbg: os.environ["FDG_ISA_DEBUG"] = dbg
    else: os.environ.pop("FDG_ISA_DEBUG", None)
    f = fd.compile_table(t, specialize="isa")
    h = f.handle
    tm = timed(lambda: h.eval_device_tiled(leaf_t.data_ptr(), 1, 64, 64 * L, root_t.data_ptr(), 1, 64, 64 * R, B, st))
    lm = timed(lambda: h.eval_device(leaf_c.data_ptr(), 1, B, root_c.data_ptr(), 1, B, B, st))
    torch.cuda.synchronize()
    same = None
    if ref is None: ref = root_c.clone()
    else: same = bool(torch.equal(ref, root_c))
    fr = lambda ms: 8 * (L + R) * B / ms / 1e6 / 8000
    print(f"FDG_ISA_DEBUG={dbg or '-':10s} tile-major {tm:.3f} ms frac {fr(tm):.3f} | leaf-major {lm:.3f} ms frac {fr(lm):.3f} | roots equal to the first run: {same}", flush=True)
