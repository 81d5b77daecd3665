"""Are the device entry points capturable in a hipGraph?  After one warm-up call they only launch kernels on the caller's
stream (no allocation, no synchronisation), so a Monte-Carlo loop of small batches can be captured once and replayed:
this script captures 20 calls of fdg_mc_accumulate_device / fdg_accumulate_device in a torch.cuda.CUDAGraph (= hipGraph on
ROCm), checks the replayed result against the eager one and times both (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "gv_sigma4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
NCALL = 20
t, z = workloads.get(name), workloads.leafstates(name)
dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"])
kF, beta, lam = 1.919, 3.0, 1.2
K = torch.rand((NCALL, n_loop * dim, B), dtype=torch.float64, device=dev) * 4 - 2
T = torch.rand((NCALL, n_tau, B), dtype=torch.float64, device=dev) * beta
w = torch.rand(B, dtype=torch.float64, device=dev)
tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
f = fd.compile_table(t, specialize="isa"); h = f.handle; h.specialize_fused(tab)
acc = torch.zeros(t.n_root, dtype=torch.float64, device=dev)
def loop():
    st = torch.cuda.current_stream().cuda_stream
    for i in range(NCALL):
        h.mc_accumulate_device(K[i].data_ptr(), 1, B, T[i].data_ptr(), 1, B, kF, beta, lam, w.data_ptr(), acc.data_ptr(), B, st)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    loop(); torch.cuda.synchronize()          # warm-up: modules loaded, workspace allocated
    acc.zero_(); loop(); torch.cuda.synchronize(); eager = acc.clone()
    g = torch.cuda.CUDAGraph()
    acc.zero_()
    with torch.cuda.graph(g, stream=s):
        loop()
    acc.zero_(); g.replay(); torch.cuda.synchronize()
    print("replayed graph == eager loop:", bool(torch.equal(acc, eager)))
    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    te, tg = timeit(loop), timeit(g.replay)
    print(f"{name}, {NCALL} calls of {B} samples: eager {te*1e3/NCALL:.1f} us per call = {B*NCALL/te*1e3:.3e} samples/s; "
          f"one hipGraph replay {tg*1e3/NCALL:.1f} us per call = {B*NCALL/tg*1e3:.3e} samples/s")
