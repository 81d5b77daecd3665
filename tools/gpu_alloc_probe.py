"""GPU dev tool: does the rate of the headline launch depend on WHERE its 70 GB leaf matrix lands?  Rounds of allocate ->
fill -> time -> free in one process, with a block of a different size allocated first each round to move the placement.
Run it plain (prints ms per round) or under `rocprofv3 --pmc <TLB / stall counters> --kernel-trace --output-format csv`
(tools/prof_alloc.sh): the per-dispatch counters then line up with the per-dispatch durations.
usage: gpu_alloc_probe.py [workload] [B] [rounds]"""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t = workloads.get(name); L, R = t.n_leaf, t.n_root
f = fd.compile_table(t, specialize="isa")
shift_mb = [0, 3, 517, 1, 2051, 64, 9000, 130][:rounds] + [0] * max(0, rounds - 8)
mode = os.environ.get("PROBE_ALLOC", "torch")
for r in range(rounds):
    pad = torch.empty(max(1, shift_mb[r]) << 20, dtype=torch.uint8, device=dev)      # moves the next allocation
    if mode == "torch":
        leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    else:                                                                              # column slabs allocated one by one
        cols = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(L)]
        raise SystemExit("slab mode needs per-column base pointers: not supported by the ABI")
    root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
    st = torch.cuda.current_stream().cuda_stream
    capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 11, 0, st)
    res = []
    for lg in [int(x) for x in os.environ.get("PROBE_GROUPS", "0").split(",")]:        # log2 of the tile-walk groups, A/B on the same placement
        os.environ["FDG_ISA_TILE_GROUPS_NOW"] = str(lg)
        for _ in range(12): f(root, leaf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for _ in range(n): f(root, leaf)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append(f"G=2^{lg}: {ms:.3f} ms frac {B * 8 * (L + R) / ms / 1e6 / 8000:.3f}")
    print(f"round {r}: pad {shift_mb[r]:5d} MB  leaf @ {leaf.data_ptr():#x}  " + "  |  ".join(res), flush=True)
    del leaf, root, pad
    torch.cuda.empty_cache()
