"""GPU dev tool: time one workload's ISA kernel under several environment settings (FDG_* knobs of the back end),
checking each against the oracle first.  python tools/gpu_env_sweep.py WORKLOAD "A=1,B=2" "A=3" ...   ("-" = no setting)"""
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
import numpy as np, torch
import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
dev = torch.device("cuda:0")
name = sys.argv[1]
t = workloads.get(name)
st = t.stats()
L, R = t.n_leaf, t.n_root
B = int(os.environ.get("SWEEP_B", max(1 << 14, min(4_000_000, int(2.4e9 / (8 * L))))))
TILED = os.environ.get("SWEEP_LAYOUT") == "tile_major"      # fdg_eval_device_tiled: [tile, leaf, sample in tile]
if TILED:
    B = (B + 63) // 64 * 64
    leaf = torch.empty((B // 64, L, 64), dtype=torch.float64, device=dev)
    root = torch.empty((B // 64, R, 64), dtype=torch.float64, device=dev)
elif os.environ.get("SWEEP_LAYOUT") == "sample_major":      # compile_Python's row-major [B, L] / [B, R]
    leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
    root = torch.empty((B, R), dtype=torch.float64, device=dev)
else:
    leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
if TILED:
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 11, 0, torch.cuda.current_stream().cuda_stream)
else:
    capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 11, 0, torch.cuda.current_stream().cuda_stream)
nchk = 4099
rows = lambda x, n: (x[:(n + 63) // 64].permute(0, 2, 1).reshape(-1, x.shape[1])[:n] if TILED else x[:n]).cpu().numpy()
run = lambda f: f.eval_tiled(root, leaf, B) if TILED else f(root, leaf)
want = oracle.eval_static(t, rows(leaf, nchk), np.zeros((nchk, R)))
for setting in sys.argv[2:]:
    kv = {} if setting == "-" else dict(x.split("=") for x in setting.split(","))
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        t0 = time.time()
        f = fd.compile_table(t, specialize="isa", cache_dir=os.environ.get("SWEEP_CACHE", "/tmp/fdg-sweep-cache"))
        tc = time.time() - t0
        root.zero_()
        run(f); torch.cuda.synchronize()
        ok = np.array_equal(rows(root, nchk), want)
        for _ in range(3): run(f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = int(os.environ.get("SWEEP_N", 10))
        e0.record()
        for _ in range(n): run(f)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        i = f.info()
        print(f"{name} [{setting}] {'exact' if ok else 'MISMATCH'} B={B} {ms:.3f} ms {B / ms * 1e3:.3e} evals/s  alg {B / ms * 1e3 * st['bytes_alg'] / 1e9:.0f} GB/s "
              f"vgpr={i['spec_vgpr']} lds={i['spec_lds_bytes']} compile {tc:.1f}s", flush=True)
        del f
    except Exception as e:
        print(f"{name} [{setting}] FAILED: {e}", flush=True)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
