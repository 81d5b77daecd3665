#!/usr/bin/env python
"""Bench of the graph-evaluator hot path (contract: see the task statement).

A "step" is one pass of the evaluator over one batch of synthetic leaf values
already resident in HBM: ``fdg_eval_device`` on B samples.  Metric:
graph-evaluations/sec, whole job.  N > 1 (launched with torch.distributed.run):
samples shard across ranks (weak scaling, no data-path collective); after the
timed steps every rank accumulates its roots and ONE all-reduce (RCCL) combines
the observable -- that reduce is inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # AMD spec, FMA counted as 2 (not in the local guide)


WORKLOAD_NOTES = {
    "sigma2": " (config 2: optimized 2-loop Parquet self-energy, transcribed from the reference's own rendering assets/sigma_o2.svg)",
    "sigma4_standin": " (config 3 stand-in (ii): seeded parquet-recursion synthetic graph with SURVEY 8d's sizes, ~10^4 nodes; the real graph needs the Julia front end)",
    "gv_sigma4": " (the 4-loop self-energy in the reference's GV form, catalog Sigma4_0_0.diag through the restated reader + optimize!)",
    "gv_sigma4_taylor2": " (config 4: 4-loop self-energy with Taylor-mode AD counterterms of order 2 in the coupling: reference GV catalog Sigma4_0_0.diag through the restated reader, taylorAD and optimize!; 7373 nodes; the 4-loop Parquet graph itself needs the Julia front end)",
    "gv_sigma5": " (config 3 stand-in (i) / config 5: reference GV catalog Sigma5_0_0.diag through the restated reader + optimize!)",
    "gv_sigma6": " (config 3 stand-in (i): reference GV catalog Sigma6_0_0.diag through the restated reader + optimize!)",
    "parquet_sigma4": " (config 3: the 4-loop Parquet self-energy, Parquet.build(DiagPara(type=SigmaDiag, innerLoopNum=4, hasTau=true, filter=[NoHartree])) + optimize! "
                      "through the restated front end (feynmandiagram.jl_amd/parquet.py; pinned by the reference's diagram counts 1, 3, 18, 171 and its rendering of the "
                      "2-loop graph); the reference's default interaction, ChargeCharge Instant: 1325 nodes after optimize!, 3646 before)",
    "parquet_sigma4_dyn": " (config 3 with a Dynamic interaction: 4819 nodes)",
    "parquet_sigma4_insdyn": " (config 3 with an Instant + Dynamic interaction: 20147 nodes)",
    "parquet_sigma4_taylor2": " (config 4: the 4-loop Parquet self-energy with Taylor-mode AD counterterms of order 2 in the coupling, restated taylorAD + optimize!: 7421 nodes, 12 roots)",
    "parquet_sigma2": " (configs 1-2 from the restated Parquet front end, one optimize! pass: 19 nodes)",
    "parquet_sigma5": " (the 5-loop Parquet self-energy: 11407 nodes; its sub-vertices use the fully irreducible vertex of the reference's GV catalogs)",
    "parquet_ver4_4": " (the graph example/benchmark.jl builds: Parquet.vertex4(DiagPara(type=Ver4Diag, innerLoopNum=4)) + optimize!, 44854 nodes, 180 roots)",
    "gv_ver4_4": " (the graph example/benchmark_GV.jl:23 builds: GV.diagsGV_ver4(4) + optimize!, catalog Vertex44_0_0.diag, 31803 nodes, 26 roots)",
}
DEFAULT_B = {"sigma2": 64_000_000, "sigma4_standin": 2_000_000, "sigma4_worstcase": 1_000_000, "synthetic_small": 8_000_000,
             "gv_sigma4": 8_000_000, "gv_sigma5": 2_000_000, "gv_sigma6": 500_000, "gv_sigma4_taylor2": 4_000_000,
             "gv_sigma5_taylor2": 1_000_000, "parquet_sigma2": 64_000_000, "parquet_sigma3": 16_000_000, "parquet_sigma4": 100_000_000,
             "parquet_sigma4_dyn": 8_000_000, "parquet_sigma4_insdyn": 4_000_000, "parquet_sigma4_taylor2": 8_000_000,
             "parquet_sigma4_dyn_taylor2": 2_000_000, "parquet_sigma4_insdyn_taylor2": 1_000_000, "parquet_sigma5": 2_000_000,
             "parquet_ver4_4": 500_000, "gv_ver4_4": 500_000}
PARITY_NOTE = ("bit-exact vs our restatement of the Julia evaluator (oracle/); the reference's known-answer tests pin structure, "
               "leaf numbering and factors, not the rounding of the n-ary folds")


class Case:
    """One workload resident on the device: handle, leaf batch (synthetic, Philox keyed by the global sample index), root buffer."""

    def __init__(self, workload, layout, B, dev, backend="isa", flags=0, sample_offset=0):
        import torch
        import feynmandiagram_jl_amd as fd
        from feynmandiagram_jl_amd import capi, workloads
        self.workload, self.layout, self.B, self.dev = workload, layout, B, dev
        self.t = t = workloads.get(workload)
        self.st = t.stats()
        L, R = t.n_leaf, t.n_root
        self.f = fd.compile_table(t, specialize={"isa": "isa", "isa-autotune": "isa-autotune", "auto": "auto", "hip": True, "interp": False}[backend], flags=flags)
        if layout == "sample_major":          # compile_Python's row-major [B, L] / [B, R]
            self.leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
            self.root = torch.empty((B, R), dtype=torch.float64, device=dev)
        else:                                 # Julia column-major B x L / B x R matrices
            self.leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
            self.root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
        self.stream = torch.cuda.current_stream()
        capi.fill_uniform_device(self.leaf.data_ptr(), B, L, self.leaf.stride(0), self.leaf.stride(1), 1234, sample_offset, self.stream.cuda_stream)

    def step(self):
        self.f(self.root, self.leaf)

    def timed(self, steps, warm):
        """`warm` untimed launches, then `steps` launches bracketed by HIP events on the launch stream.  Returns ms per launch (list)."""
        import torch
        for _ in range(warm):
            self.step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record(self.stream)
        for i in range(steps):
            self.step()
            ev[i + 1].record(self.stream)
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]

    def parity_sample(self, n=2048):
        """The first n samples of the last launch against the oracle's restatement of the compiled evaluator (C, one thread)."""
        import numpy as np
        import oracle
        n = int(min(n, self.B))
        want = oracle.eval_static(self.t, np.ascontiguousarray(self.leaf[:n].cpu().numpy()), np.zeros((n, self.t.n_root)))
        got = self.root[:n].cpu().numpy()
        return bool(np.array_equal(got, want)), float(np.abs(got - want).max()) if n else 0.0, n


def observable_sum(root):
    """Column sums of the root matrix, the observable of the final reduction.  For a Julia-layout matrix (R long rows)
    torch's reduction runs one workgroup per row -- 50 ms for 4 x 10^8 doubles -- so the rows are summed in two stages."""
    if root.stride(0) != 1 or root.shape[0] < (1 << 16):
        return root.sum(dim=0)
    rt = root.t()                                  # [R, B], rows contiguous
    R, B = rt.shape
    c = 1 << 14
    nb = B // c
    acc = rt[:, :nb * c].reshape(R, nb, c).sum(dim=2).sum(dim=1)
    return acc + rt[:, nb * c:].sum(dim=1) if nb * c < B else acc


def roofline_of(st, B, avg_kernel_s, kernel, accumulate=False):
    bytes_per_eval = st["bytes_alg_accumulate"] if accumulate else st["bytes_alg"]
    achieved = bytes_per_eval * B / avg_kernel_s / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": kernel, "avg_kernel_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_eval * B}


def attach_traffic(roof, workload, layout, B, avg_kernel_s):
    """HBM bytes per launch from the rocprofv3 --pmc passes of the same command (bench.py cannot collect counters on
    itself): profiles/r02_traffic.json holds bytes per evaluation, scaled here to this batch -- a value from the named
    profile, not a measurement of this run."""
    for fn in ("r02_traffic.json", "r01_traffic.json"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", fn))).get(workload if layout == "leaf_major" else workload + ":" + layout)
        except (OSError, ValueError):
            continue
        if tr and tr.get("layout", "leaf_major") == layout:
            roof["traffic"] = tr["bytes_per_eval"] * B
            roof["traffic_gbs"] = tr["bytes_per_eval"] * B / avg_kernel_s / 1e9
            roof["traffic_frac"] = roof["traffic_gbs"] / HBM_PEAK_GBS
            roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
            roof["traffic_source"] = ("NOT measured in this run: (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes, profiles/" + fn +
                                      " (" + tr.get("source", "") + "), per evaluation, scaled to this batch")
            return


def measured_copy(dev):
    """The box's own streaming ceiling: a 2 GiB device-to-device copy by fdg_copy_device (16 bytes per lane), read + write counted."""
    import torch
    from feynmandiagram_jl_amd import capi
    a = torch.empty(1 << 28, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        capi.copy_device(b.data_ptr(), a.data_ptr(), a.numel(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        capi.copy_device(b.data_ptr(), a.data_ptr(), a.numel(), st)
    e1.record()
    torch.cuda.synchronize()
    return 10 * 2 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def secondary_case(workload, layout, dev, steps=20, warm=30, copy_gbs=None):
    """A workload outside the headline, measured in the same process: `warm` untimed + `steps` timed launches, its own
    roofline fraction, and a bitwise check of a sample against the oracle."""
    import torch
    try:
        c = Case(workload, layout, 16_000_000 if workload == "parquet_sigma4" else DEFAULT_B.get(workload, 1_000_000), dev)
        ms = c.timed(steps, warm)
        avg = sum(ms) / len(ms) / 1e3
        ok, dev_max, n = c.parity_sample()
        kern = ("fdg_isa_eval_coop" if workload in ("sigma4_standin", "sigma4_worstcase") else "fdg_isa_eval_nt" if layout == "leaf_major" and c.B % 16 == 0
                else "fdg_isa_eval_rm" if layout == "sample_major" else "fdg_isa_eval")
        roof = roofline_of(c.st, c.B, avg, kern)
        attach_traffic(roof, workload, layout, c.B, avg)
        if copy_gbs:
            roof["frac_of_measured_copy"] = roof["achieved"] / copy_gbs
        info = c.f.info()
        out = {"workload": workload + WORKLOAD_NOTES.get(workload, ""), "layout": layout, "value": c.B / avg, "unit": "evals/s",
               "samples_per_launch": c.B, "timed_launches": steps, "warmup_launches": warm, "avg_kernel_ms": avg * 1e3,
               "n_leaf": c.t.n_leaf, "n_node": c.t.n_node, "n_root": c.t.n_root, "flops_per_eval": c.st["flops_alg"], "bytes_per_eval": c.st["bytes_alg"],
               "roofline": roof,
               "valu_fp64_tflops": c.st["flops_alg"] * c.B / avg / 1e12,
               "kernel_info": {k: info[k] for k in ("max_live", "spec_vgpr", "spec_lds_bytes")},
               "gpu_matches_cpu_bitwise": ok, "max_abs_dev": dev_max, "parity_samples": n}
        del c
        torch.cuda.empty_cache()
        return out
    except Exception as e:                      # secondary: never takes the headline line down
        return {"workload": workload, "layout": layout, "error": f"{type(e).__name__}: {e}"}


def config5(dev, rank, world, dist, comm, steps, warm):
    """BASELINE.json config 5 (example/benchmark_GV.jl as BASELINE.json words it: the GV 5th-order self-energy, samples
    sharded over the GPUs, one final reduce): per step fdg_accumulate_device on this rank's shard -- weighted
    accumulation inside the evaluator, roots never reach HBM --, after the last step ONE all-reduce of R doubles."""
    import torch
    from feynmandiagram_jl_amd.sharding import reduce_observable, shard_range
    try:
        B = DEFAULT_B["gv_sigma5"]
        start, count = shard_range(B * world, rank, world)
        c = Case("gv_sigma5", "leaf_major", count, dev, sample_offset=start)
        w = torch.rand(count, dtype=torch.float64, device=dev)
        acc = torch.zeros(c.t.n_root, dtype=torch.float64, device=dev)
        for _ in range(warm):
            c.f.accumulate(c.leaf, w, acc)
        acc.zero_()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record(c.stream)
        for i in range(steps):
            c.f.accumulate(c.leaf, w, acc)
            ev[i + 1].record(c.stream)
        reduce_observable(acc, comm=comm)     # the one collective: R doubles over xGMI (RCCL)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
        avg = sum(ms) / len(ms) / 1e3
        total = float(count) * steps * world
        roof = roofline_of(c.st, count, avg, "fdg_isa_eval_acc + fdg_reduce_lane_partials", accumulate=True)
        out = {"workload": "gv_sigma5" + WORKLOAD_NOTES["gv_sigma5"], "value": total / elapsed, "unit": "samples/s (whole job)", "n_gpus": world,
               "steps": steps, "warmup": warm, "samples_per_step_per_gpu": count, "total_samples": total,
               "samples_to_config5_total": "1e9 samples = %d steps of this size at this GPU count" % round(1e9 / (count * world)),
               "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "roofline_rank0": roof,
               "observable": [float(x) for x in acc.cpu()],
               "what": "fdg_accumulate_device per step on the rank's shard; one all-reduce of R doubles after the last step, inside the timed region"}
        del c, w
        torch.cuda.empty_cache()
        return out
    except Exception as e:
        return {"workload": "gv_sigma5", "error": f"{type(e).__name__}: {e}"}


def main():
    # The result line must be the only thing on stdout.  RCCL writes a version banner to the C-level stdout
    # (buffered, so it would land AFTER our line at exit); everything else this process or its libraries print
    # goes to stderr, and the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # one step = config 3's 1e8 samples on the default workload
    ap.add_argument("--warmup", type=int, default=60)   # power management needs ~40 launches (50 ms) to settle: 1.4 -> 1.04 ms per launch
    ap.add_argument("--workload", default="parquet_sigma4")
    ap.add_argument("--samples", type=int, default=0, help="samples per step per GPU (0 = workload default)")
    ap.add_argument("--layout", default="leaf_major", choices=["sample_major", "leaf_major"],
                    help="leaf_major = a Julia column-major B x L matrix (the host language's native layout); "
                         "sample_major = compile_Python's row-major [B, L]")
    ap.add_argument("--backend", default="isa", choices=["isa", "isa-autotune", "auto", "hip", "interp"],
                    help="isa: optimizing back end, gfx950 assembly; hip: straight-line HIP source via hiprtc; interp: table interpreter")
    ap.add_argument("--interp", action="store_true", help="same as --backend interp")
    ap.add_argument("--comm", default="torch", choices=["torch", "fdg"],
                    help="who runs the one reduction of the observable: torch.distributed (nccl == RCCL) or libfdg's fdg_comm_* (RCCL)")
    ap.add_argument("--fast-math", action="store_true",
                    help="FDG_SPEC_FAST_MATH: fused multiply-adds; within 1e-12 of the term scale but NOT bit-identical to the reference "
                         "(reported separately, never the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc-step", action="store_true", help="skip the secondary measurement of the whole Monte-Carlo step (leaves from momenta and times)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other workloads (config 2, 3 stand-ins, 5, row-major layout) measured after the headline")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import feynmandiagram_jl_amd as fd
    from feynmandiagram_jl_amd import capi, workloads
    from feynmandiagram_jl_amd.sharding import make_comm, reduce_observable, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or os.environ.get("FDG_BENCH_FORCE_DIST"):   # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # nccl == RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = make_comm(rank, world) if (dist and args.comm == "fdg") else None

    if args.interp:
        args.backend = "interp"
    # Samples resident per step.  Decimal sizes on purpose: BASELINE.json's sample counts are decimal (25 steps of the
    # default = config 3's 10^8 samples), and a leaf-major matrix whose column stride is a power of two aliases HBM
    # channels (measured: -3 % on the default workload, -14 % on sigma2).
    B = args.samples or DEFAULT_B.get(args.workload, 1_000_000)
    t_probe = workloads.get(args.workload)
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    while not args.samples and 8 * B * (t_probe.n_leaf + t_probe.n_root) > 0.6 * free_b and B > 1_000_000:
        B //= 2                         # (a 288 GB device holds config 3's 1e8 samples of 84 leaves, 70 GB, with room to spare)
    # per-rank Philox offset: results do not depend on how samples are sharded
    start, count = shard_range(B * world, rank, world)          # weak scaling: B samples per GPU
    assert count == B
    case = Case(args.workload, args.layout, B, dev, backend=args.backend, flags=capi.FDG_SPEC_FAST_MATH if args.fast_math else 0,
                sample_offset=start)
    if args.backend == "isa-autotune":
        args.backend = "isa"
    t, st, f, leaf, root, stream = case.t, case.st, case.f, case.leaf, case.root, case.stream
    L, R = t.n_leaf, t.n_root
    step = case.step

    step()
    _ = observable_sum(root)              # load the reduction used for the final observable now: a pause between the
    torch.cuda.synchronize()              # warm-up and the timed steps would let the clocks fall back
    # Clock settling: after idle the first ~50 launches run at transient clocks (boost, then throttle, then the
    # sustained state: 1.10 -> 1.40 -> 1.05-1.15 ms per launch on the default workload).  The timed steps are meant to
    # show the sustained rate, so at least 60 untimed launches precede them whatever --warmup says (disclosed in the
    # JSON line as config.settle_steps; they are the same step as the warm-up and the timed ones).
    settle = max(0, 60 - args.warmup)
    for _ in range(settle + args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        step()
        ev[i + 1].record(stream)          # same stream the kernel is launched on
    acc = observable_sum(root)            # final observable accumulation
    reduce_observable(acc, comm=comm)     # the one collective: R doubles over xGMI (RCCL)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    avg_kernel_s = sum(kern_ms) / len(kern_ms) / 1e3

    total_evals = float(B) * args.steps * world
    value = total_evals / elapsed
    info = f.info()
    out = {
        "metric": "graph-evaluations/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if not args.fast_math else "f64 (fused multiply-add: within 1e-12, not bit-identical)",
        "data": "synthetic",
        "config": {"workload": args.workload + WORKLOAD_NOTES.get(args.workload, ""),
                   "graph": t.name, "n_leaf": L, "n_node": t.n_node, "n_edge": t.n_edge, "n_root": R,
                   "flops_per_eval": st["flops_alg"], "bytes_per_eval": st["bytes_alg"],
                   "samples_per_step_per_gpu": B, "layout": args.layout, "settle_steps": settle,
                   "kernel": {"isa": "fdg_isa_eval (per-graph gfx950 assembly)", "hip": "fdg_spec (per-graph HIP source, hiprtc)",
                              "auto": "fdg_isa_eval, or its HIP-source companion fdg_spec_sm for row-major input of small graphs",
                              "interp": "fdg_interp (table interpreter)"}[args.backend],
                   "parallelism": f"samples sharded x{world}, one all-reduce of {R} doubles",
                   "parity": PARITY_NOTE},
    }
    kname = {"isa": "fdg_isa_eval", "interp": "fdg_interp", "auto": "fdg_isa_eval / fdg_spec_sm",
             "hip": "fdg_spec_sm" if args.layout == "sample_major" else "fdg_spec_gen"}[args.backend]
    if args.backend == "isa" and args.layout == "leaf_major" and B % 16 == 0 and not os.environ.get("FDG_ISA_NO_STREAMING"):
        kname = "fdg_isa_eval_nt"        # line-aligned column-major batch: the streaming form of the kernel
    copy_gbs = None
    if rank == 0:
        out["roofline"] = roofline_of(st, B, avg_kernel_s, kname)
        achieved = out["roofline"]["achieved"]
        try:
            copy_gbs = measured_copy(dev)
            out["roofline"]["measured_copy_gbs"] = copy_gbs
            out["roofline"]["measured_copy_kernel"] = "fdg_copy_device (16 B per lane, non-temporal loads and stores, 2 GiB, read + write counted)"
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
        except RuntimeError:
            pass
        if args.backend == "isa":
            attach_traffic(out["roofline"], args.workload, args.layout, B, avg_kernel_s)
        out["valu_fp64"] = {"achieved_tflops": st["flops_alg"] * B / avg_kernel_s / 1e12,
                            "peak_tflops_fma": FP64_VALU_PEAK_TFLOPS,
                            "note": "secondary ceiling: add/mul only (no FMA contraction allowed), so the usable peak is half"}
        out["kernel_info"] = {k: info[k] for k in ("max_live", "spec_vgpr", "spec_lds_bytes", "spec_scratch_bytes")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(t, leaf, root, args.cpu_seconds)
    del case, leaf, root, f, step
    torch.cuda.empty_cache()
    # ---- after and outside the headline's timed region ---------------------------------------------------------
    if args.backend == "isa" and not args.no_secondary and not args.fast_math:
        # config 5 runs on every rank (its one collective needs them all); the rest on rank 0 at N = 1 only
        c5 = config5(dev, rank, world, dist if world > 1 or os.environ.get("FDG_BENCH_FORCE_DIST") else None, comm, steps=max(20, min(args.steps, 100)), warm=30)
        if rank == 0:
            out["config5"] = c5
        if rank == 0 and world == 1:
            sec = []
            head = (args.workload, args.layout)
            for wl, lay in (("parquet_sigma4", "leaf_major"), ("parquet_sigma4", "sample_major"), ("parquet_sigma4_dyn", "leaf_major"),
                            ("parquet_sigma4_insdyn", "leaf_major"), ("parquet_sigma4_taylor2", "leaf_major"), ("parquet_sigma5", "leaf_major"),
                            ("parquet_ver4_4", "leaf_major"), ("gv_ver4_4", "leaf_major"), ("sigma2", "leaf_major"), ("sigma4_standin", "leaf_major"), ("gv_sigma4", "leaf_major"), ("gv_sigma5", "leaf_major"),
                            ("gv_sigma6", "leaf_major"), ("gv_sigma4_taylor2", "leaf_major"), ("gv_sigma4_taylor2", "sample_major")):
                if (wl, lay) != head:
                    sec.append(secondary_case(wl, lay, dev, copy_gbs=copy_gbs))
            out["secondary"] = sec
            out["secondary_note"] = ("measured in this process after the headline's timed region (30 untimed + 20 timed launches each, HIP events on the "
                                     "launch stream); never part of `value`.  " + PARITY_NOTE)
    if rank == 0:
        if world == 1 and not args.no_mc_step and args.backend == "isa":
            out["mc_step"] = mc_step(t, args.workload, min(B, 16_000_000), dev, args.fast_math)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()


def mc_step(t, workload, B, dev, fast_math):
    """Secondary figure, outside the timed region and not part of `value`: the whole Monte-Carlo integrand step of
    example/benchmark.jl:58-87 on the same graph -- leaves computed from the sample's loop momenta and times, graph,
    weighted accumulation -- through fdg_graph_specialize_fused / fdg_mc_accumulate_device.  Needs the
    graph's leafstates tables (feynmandiagram.jl_amd/data/, derived from the reference's GV catalogs)."""
    import numpy as np
    import torch
    import feynmandiagram_jl_amd as fd
    from feynmandiagram_jl_amd import capi
    from feynmandiagram_jl_amd import workloads
    try:
        z = workloads.leafstates(workload)
        if z is None:
            return None
        dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"])
        kF, beta, lam = 1.919, 3.0, 1.2
        dK = torch.rand((n_loop * dim, B), dtype=torch.float64, device=dev) * 4 - 2        # component-major, like a Julia B x n matrix
        dT = torch.rand((n_tau, B), dtype=torch.float64, device=dev) * beta
        w = torch.rand(B, dtype=torch.float64, device=dev)
        acc = torch.zeros(t.n_root, dtype=torch.float64, device=dev)
        tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau,
                                           kF, beta, lam)
        h = fd.compile_table(t, specialize="isa", flags=capi.FDG_SPEC_FAST_MATH if fast_math else 0).handle
        h.specialize_fused(tab)
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: h.mc_accumulate_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, w.data_ptr(), acc.data_ptr(), B, st)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        return {"value": B / ms * 1e3, "unit": "samples/s", "ms_per_call": ms, "samples_per_call": B,
                "what": "fdg_mc_accumulate_device: leaves from (K, T) + graph + weighted accumulation; on this handle one kernel of the "
                        "optimizing back end for programs of up to 40 000 + 30 L ops (leaves are values computed in registers), leaf kernel + evaluator above",
                "input_bytes_per_sample": 8 * (n_loop * dim + n_tau + 1), "parameters": {"kF": kF, "beta": beta, "lambda": lam},
                "parity": "graph part bit-exact given the kernel's leaves; leaves within 1e-13 relative / 1e-12 of the largest Leibniz term of the oracle's (tests/test_gpu_parity.py)"}
    except Exception as e:                      # secondary: never takes the headline line down
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline(t, leaf, root, budget_s):
    """The reference's own C back-end text (to_Cstr shape, static.jl:155-197)
    compiled by gcc -O2 -ffp-contract=off (-O1 above 2*10^4 nodes, where -O2
    needs minutes) and called once per sample on the host cores, on a bounded
    sample of the same leaf data; also checks the GPU roots of that sample."""
    import numpy as np
    import oracle
    from feynmandiagram_jl_amd.lowering import table_to_Cstr
    cores = os.cpu_count() or 1
    opt = "-O2" if t.n_node <= 20000 else "-O1"
    cb = oracle.CBaseline(table_to_Cstr(t), t.n_leaf, t.n_root, opt=opt)
    # single-core rate first (also the calibration for the threaded run)
    n1 = min(int(leaf.shape[0]), 4096)
    h1 = np.ascontiguousarray(leaf[:n1].cpu().numpy())
    cb(h1[:256], 1)
    t0 = time.perf_counter()
    cb(h1, 1)
    rate1 = n1 / max(time.perf_counter() - t0, 1e-9)
    want = max(4096, rate1 * cores * 1.0)
    n = int(min(leaf.shape[0], want, 1 << 22))
    h = np.ascontiguousarray(leaf[:n].cpu().numpy())
    cb(h[: min(n, 64 * cores)], cores)           # thread start-up outside the clock
    t0 = time.perf_counter()
    cb(h, cores)                                 # calibration pass with all threads
    t_pass = max(time.perf_counter() - t0, 1e-6)
    reps = int(max(1, min(10000, round(budget_s / t_pass))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = cb(h, cores)
    dt = time.perf_counter() - t0
    got = root[:n].cpu().numpy()
    # the interpreter the reference's examples actually call (IR.eval!, example/benchmark.jl:84-86):
    # its arithmetic restated in oracle/fdg_oracle.c, one thread
    ni = int(min(n, 2048))
    t0 = time.perf_counter()
    ri = oracle.eval_interp(t, h[:ni])
    dti = max(time.perf_counter() - t0, 1e-9)
    return {"value": n * reps / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{reps} pass(es) over {n} samples of the same leaf batch, {dt:.2f} s on {cores} threads; reference's to_Cstr text compiled by gcc {opt} -ffp-contract=off (the Julia evaluator cannot run here)",
            "single_core_evals_per_s": rate1, "gcc_compile_s": cb.compile_seconds,
            "eval_interp_single_core_evals_per_s": ni / dti,
            "eval_interp_max_rel_dev_vs_compiled": float(np.max(np.abs(ri - ref[:ni]) / np.maximum(np.abs(ref[:ni]), 1e-300))),
            "gpu_matches_cpu_bitwise": bool(np.array_equal(got, ref)),
            "max_abs_dev": float(np.abs(got - ref).max())}


if __name__ == "__main__":
    main()
