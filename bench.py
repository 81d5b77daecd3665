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


def main():
    # The result line must be the only thing on stdout.  RCCL writes a version banner to the C-level stdout
    # (buffered, so it would land AFTER our line at exit); everything else this process or its libraries print
    # goes to stderr, and the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 steps x 4e6 samples = 4e8 samples (config 3 names 1e8)
    ap.add_argument("--warmup", type=int, default=60)   # power management needs ~40 launches (50 ms) to settle: 1.4 -> 1.04 ms per launch
    ap.add_argument("--workload", default="gv_sigma4_taylor2")
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

    t = workloads.get(args.workload)
    st = t.stats()
    L, R = t.n_leaf, t.n_root
    # Samples resident per step.  Decimal sizes on purpose: BASELINE.json's sample counts are decimal (25 steps of the
    # default = config 3's 10^8 samples), and a leaf-major matrix whose column stride is a power of two aliases HBM
    # channels (measured: -3 % on the default workload, -14 % on sigma2).
    default_B = {"sigma2": 64_000_000, "sigma4_standin": 2_000_000, "sigma4_worstcase": 1_000_000, "synthetic_small": 8_000_000,
                 "gv_sigma4": 8_000_000, "gv_sigma5": 2_000_000, "gv_sigma6": 500_000, "gv_sigma4_taylor2": 4_000_000,
                 "gv_sigma5_taylor2": 1_000_000}.get(args.workload, 1_000_000)
    B = args.samples or default_B
    if args.interp:
        args.backend = "interp"
    f = fd.compile_table(t, specialize={"isa": "isa", "isa-autotune": "isa-autotune", "auto": "auto", "hip": True, "interp": False}[args.backend],
                         flags=capi.FDG_SPEC_FAST_MATH if args.fast_math else 0)
    if args.backend == "isa-autotune":
        args.backend = "isa"

    if args.layout == "sample_major":
        leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
    else:
        leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
    if args.layout == "sample_major":
        root = torch.empty((B, R), dtype=torch.float64, device=dev)          # compile_Python's row-major [B, R]
    else:
        # a Julia column-major B x R matrix, like the leaves: a wave's 64 values of one root are one 512-byte line-aligned
        # store (row-major roots leave L2 as partial lines: 2.5x the bytes for R = 6; tools/gpu_root_layout.py)
        root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
    stream = torch.cuda.current_stream()
    # per-rank Philox offset: results do not depend on how samples are sharded
    start, count = shard_range(B * world, rank, world)          # weak scaling: B samples per GPU
    assert count == B
    capi.fill_uniform_device(leaf.data_ptr(), B, L, leaf.stride(0), leaf.stride(1), 1234, start, stream.cuda_stream)

    def step():
        f(root, leaf)

    step()
    _ = root.sum(dim=0)                   # load the reduction used for the final observable now: a pause between the
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
    acc = root.sum(dim=0)                 # final observable accumulation
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
        "config": {"workload": args.workload + {"sigma4_standin": " (seeded parquet-recursion stand-in for the 4-loop Parquet self-energy, ~10^4 nodes; the real graph needs the Julia front end)",
                                                "gv_sigma4_taylor2": " (4-loop self-energy with Taylor-mode AD counterterms of order 2 in the coupling: reference GV catalog Sigma4_0_0.diag through the restated reader, taylorAD and optimize!; 7373 nodes; the 4-loop Parquet graph itself needs the Julia front end)",
                                                "gv_sigma5": " (reference GV catalog Sigma5_0_0.diag through the restated reader + optimize!)",
                                                "gv_sigma6": " (reference GV catalog Sigma6_0_0.diag through the restated reader + optimize!)"}.get(args.workload, ""),
                   "graph": t.name, "n_leaf": L, "n_node": t.n_node, "n_edge": t.n_edge, "n_root": R,
                   "flops_per_eval": st["flops_alg"], "bytes_per_eval": st["bytes_alg"],
                   "samples_per_step_per_gpu": B, "layout": args.layout, "settle_steps": settle,
                   "kernel": {"isa": "fdg_isa_eval (per-graph gfx950 assembly)", "hip": "fdg_spec (per-graph HIP source, hiprtc)",
                              "auto": "fdg_isa_eval, or its HIP-source companion fdg_spec_sm for row-major input of small graphs",
                              "interp": "fdg_interp (table interpreter)"}[args.backend],
                   "parallelism": f"samples sharded x{world}, one all-reduce of {R} doubles"},
    }
    if rank == 0:
        bytes_per_launch = st["bytes_alg"] * B
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "kernel": {"isa": "fdg_isa_eval", "interp": "fdg_interp", "auto": "fdg_isa_eval / fdg_spec_sm",
                                      "hip": "fdg_spec_sm" if args.layout == "sample_major" else "fdg_spec_gen"}[args.backend],
                           "avg_kernel_ms": avg_kernel_s * 1e3,
                           "algorithmic_bytes_per_launch": bytes_per_launch}
        # the box's own streaming ceiling next to the 8 TB/s spec: a 2 GiB device-to-device copy,
        # read + write counted, outside the timed region
        try:
            a = torch.empty(1 << 28, dtype=torch.float64, device=dev)
            b = torch.empty_like(a)
            for _ in range(2):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["roofline"]["measured_copy_gbs"] = copy_gbs
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            del a, b
        except RuntimeError:
            pass
        # HBM traffic per launch: measured separately with rocprofv3 --pmc (bench.py cannot collect
        # counters on itself); profiles/r01_traffic.json holds bytes per evaluation for the default configs
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json"))).get(args.workload)
            if tr and tr["layout"] == args.layout and args.backend == "isa":
                out["roofline"]["traffic"] = tr["bytes_per_eval"] * B
                # the same launch time against the bytes the PMC counters saw move (re-read leaves, partial-line root writes)
                out["roofline"]["traffic_gbs"] = tr["bytes_per_eval"] * B / avg_kernel_s / 1e9
                out["roofline"]["traffic_frac"] = out["roofline"]["traffic_gbs"] / HBM_PEAK_GBS
                out["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " + tr.get("source", "profiles/") + " (per evaluation, scaled to this batch)"
        except (OSError, ValueError):
            pass
        out["valu_fp64"] = {"achieved_tflops": st["flops_alg"] * B / avg_kernel_s / 1e12,
                            "peak_tflops_fma": FP64_VALU_PEAK_TFLOPS,
                            "note": "secondary ceiling: add/mul only (no FMA contraction allowed), so the usable peak is half"}
        out["kernel_info"] = {k: info[k] for k in ("max_live", "spec_vgpr", "spec_lds_bytes", "spec_scratch_bytes")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(t, leaf, root, args.cpu_seconds)
        if world == 1 and not args.no_mc_step and args.backend == "isa":
            del leaf, root
            out["mc_step"] = mc_step(t, args.workload, B, dev, args.fast_math)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()


def mc_step(t, workload, B, dev, fast_math):
    """Secondary figure, outside the timed region and not part of `value`: the whole Monte-Carlo integrand step of
    example/benchmark.jl:58-87 on the same graph -- leaves computed from the sample's loop momenta and times, graph,
    weighted accumulation -- through fdg_graph_specialize_fused / fdg_mc_accumulate_device.  Needs the
    graph's leafstates tables (tests/golden/, derived from the reference's GV catalogs)."""
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
                        "optimizing back end for programs of up to 40 000 ops (leaves are values computed in registers), leaf kernel + evaluator above",
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
