"""Where does the time of the ISA kernel go?  Times it (a) on a real leaf-major matrix, (b) with leaf stride 0
(every leaf reads the same 512 B of a tile: same instruction stream, loads hit L2), for a few prefetch
distances.  Dev tool, run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads

dev = torch.device("cuda:0")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gv_sigma4_taylor2"]
opts = [None] + [dict(n_reg=120, n_lds=40, lookahead_leaf=la, vn_window=200) for la in (100, 300, 600)]
for name in names:
    t = workloads.get(name)
    B = 1 << 21
    leaf = torch.rand((t.n_leaf, B), dtype=torch.float64, device=dev)
    root = torch.empty((t.n_root, B), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for opt in opts:
        f = fd.compile_table(t, specialize="isa", opt=opt)
        tiled = "tiled" in os.environ.get("FDG_ISA_DEBUG", "")
        for tag, ss, ls in ((("tiled", t.n_leaf, 64),) if tiled else (("hbm", 1, B), ("ls=0", 1, 0))):
            def run():
                f.handle.eval_device(leaf.data_ptr(), ss, ls, root.data_ptr(), 1, B, B, st)
            run(); run(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): run()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            print(f"{name} opt={opt} {tag}: {best:.3f} ms {B/best*1e3:.3e} evals/s", flush=True)
