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
# The arithmetic contract forbids contraction, so a fold step is one v_add_f64 / v_mul_f64: half the FMA figure.
# 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3e12 lane-operations per second (measured: 38.0e12 at eight waves per SIMD, 36-37.7e12 at one
# or two waves of a 248-register kernel, straight-line bodies, at 2.39-2.45 GHz: tools/ubench/fp64_issue.hip, profiles/r03_ubench_fp64_issue.txt).
FP64_NOFMA_PEAK_TOPS = 39.3
# The power roof (round 5; profiles/r05_log_power_probe.txt, r05_log_power_calib.txt): every row but the smallest graph runs at the package's
# 1400 W cap with the shader clock pulled down to fit.  Socket power of the evaluator's kernels fits  idle + E_BYTE * (algorithmic bytes/s) + E_OP * (fold steps/s)
# (calibrated on two rows -- sigma2 at 1120 W, the headline at 1381 W -- and within 7 % of the other memory- and ridge-bound rows), so a graph cannot be
# evaluated faster than (cap - idle) / (E_BYTE * bytes + E_OP * ops) times a second whatever the schedule.  `frac_power` = achieved / that.
POWER_CAP_W, POWER_IDLE_W, E_BYTE_J, E_OP_J = 1400.0, 240.0, 134e-12, 25.5e-12
SPEC_CLOCK_GHZ = 2.4            # the clock the 39.3 is quoted at; roofline.clock_ghz is what the chip sustained (power budget)
FP64_NOFMA_MEASURED_TOPS = 38.0
CONFIG5_TOTAL_SAMPLES = 1_000_000_000     # BASELINE.json config 5: 10^9 samples over the GPUs of the node
LINE_LIMIT = 4000                         # bytes of the stdout line (the driver keeps the last ~8 KB of stdout)
DETAIL_PATH = os.path.join(ROOT, "bench_detail.json")


WORKLOAD_NOTES = {
    "sigma2": " (config 2: optimized 2-loop Parquet self-energy, transcribed from the reference's own rendering assets/sigma_o2.svg)",
    "sigma4_standin": " (config 3 stand-in (ii): seeded parquet-recursion synthetic graph with SURVEY 8d's sizes, ~10^4 nodes; the real graph needs the Julia front end)",
    "gv_sigma4": " (the 4-loop self-energy in the reference's GV form, catalog Sigma4_0_0.diag through the restated reader + optimize!)",
    "gv_sigma4_taylor2": " (config 4: 4-loop self-energy with Taylor-mode AD counterterms of order 2 in the coupling: reference GV catalog Sigma4_0_0.diag through the restated reader, taylorAD and optimize!; 7373 nodes; the 4-loop Parquet graph itself needs the Julia front end)",
    "gv_sigma5": " (config 3 stand-in (i) / config 5: reference GV catalog Sigma5_0_0.diag through the restated reader + optimize!)",
    "gv_sigma6": " (config 3 stand-in (i): reference GV catalog Sigma6_0_0.diag through the restated reader + optimize!)",
    "parquet_sigma4": " (config 3: the 4-loop Parquet self-energy, Parquet.build(DiagPara(type=SigmaDiag, innerLoopNum=4, hasTau=true, filter=[NoHartree])) + optimize! "
                      "through the restated front end (feynmandiagram.jl_amd/producers/parquet.py; pinned by the reference's diagram counts 1, 3, 18, 171 and its rendering of the "
                      "2-loop graph); the reference's default interaction, ChargeCharge Instant: 1325 nodes after optimize!, 3646 before)",
    "parquet_sigma4_dyn": " (config 3 with a Dynamic interaction: 4819 nodes)",
    "parquet_sigma4_insdyn": " (config 3 with an Instant + Dynamic interaction: 20147 nodes)",
    "parquet_sigma4_taylor2": " (config 4: the 4-loop Parquet self-energy with Taylor-mode AD counterterms of order 2 in the coupling, restated taylorAD + optimize!: 7421 nodes, 12 roots)",
    "parquet_sigma2": " (configs 1-2 from the restated Parquet front end, one optimize! pass: 19 nodes)",
    "parquet_sigma5": " (the 5-loop Parquet self-energy: 11407 nodes; its sub-vertices use the fully irreducible vertex of the reference's GV catalogs)",
    "parquet_ver4_4": " (the graph example/benchmark.jl builds: Parquet.vertex4(DiagPara(type=Ver4Diag, innerLoopNum=4)) + optimize!, 44854 nodes, 180 roots)",
    "gv_ver4_4": " (the graph example/benchmark_GV.jl:23 builds: GV.diagsGV_ver4(4) + optimize!, catalog Vertex44_0_0.diag, 31803 nodes, 26 roots)",
}
SHORT_NOTES = {     # <= 120 characters: what the stdout line says about the workload (the long notes go to bench_detail.json)
    "parquet_sigma4": "parquet_sigma4: config 3, 4-loop Parquet self-energy (NoHartree) + optimize!, restated front end, 1325 nodes",
    "parquet_sigma4_taylor2": "parquet_sigma4_taylor2: config 4, 4-loop Parquet self-energy + Taylor AD order 2 in the coupling, 7421 nodes",
    "gv_sigma5": "gv_sigma5: config 5, GV 5th-order self-energy (catalog Sigma5_0_0.diag) + optimize!, 3897 nodes",
    "sigma2": "sigma2: config 2, optimized 2-loop Parquet self-energy (reference rendering assets/sigma_o2.svg)",
}
DEFAULT_B = {"sigma2": 64_000_000, "sigma4_standin": 2_000_000, "sigma4_worstcase": 1_000_000, "synthetic_small": 8_000_000,
             "gv_sigma4": 8_000_000, "gv_sigma5": 2_000_000, "gv_sigma6": 500_000, "gv_sigma4_taylor2": 4_000_000,
             "gv_sigma5_taylor2": 1_000_000, "parquet_sigma2": 64_000_000, "parquet_sigma3": 16_000_000, "parquet_sigma4": 100_000_000,
             "parquet_sigma4_dyn": 8_000_000, "parquet_sigma4_insdyn": 4_000_000, "parquet_sigma4_taylor2": 8_000_000,
             "parquet_sigma4_dyn_taylor2": 2_000_000, "parquet_sigma4_insdyn_taylor2": 1_000_000, "parquet_sigma5": 2_000_000,
             "parquet_ver4_4": 512_000, "gv_ver4_4": 512_000}          # (whole 64-sample tiles: the pooled cooperative kernel takes full tiles)
PAIR_ALL = False         # --pair-all (experiment): every tile-major / row-major secondary row through fdg_batch_alloc_pair
POWER_LEG_S = 1.5        # seconds of back-to-back launches per secondary row under rocm-smi (--power-seconds; 0: off)
PAIRED_ROWS = {("parquet_sigma4", "tile_major"), ("parquet_sigma4", "leaf_major"), ("parquet_sigma4", "sample_major"), ("parquet_sigma4_dyn", "tile_major"), ("sigma2", "tile_major"),
               ("gv_sigma4", "tile_major"), ("gv_sigma4_taylor2", "tile_major")}
PARITY_NOTE = ("bit-exact vs our restatement of the Julia evaluator (oracle/); the reference's known-answer tests pin structure, "
               "leaf numbering and factors, not the rounding of the n-ary folds")


DRY = False      # --dry-run: no device work at all (gloo on CPU tensors); exercises sharding, the collective and the stdout line


def sync():
    if not DRY:
        import torch
        torch.cuda.synchronize()


def settle_after_free(nbytes):
    """The driver wipes released device memory in the background at 18-28 GB/s, and while that write stream lasts every kernel runs 2-7 % slower
    (round 5: profiles/r05_log_pair_alloc_settle.txt).  A measurement that follows the release of a large batch waits for the wipe first."""
    if not DRY and nbytes > (64 << 20):
        import torch
        torch.cuda.synchronize()
        time.sleep(nbytes / 16e9 + 0.2)


class Stamps:
    """n+1 time stamps on the launch stream: HIP events (torch.cuda.Event sees only torch's current stream -- the one
    the kernels are launched on), or host clocks in a dry run."""

    def __init__(self, n, stream):
        import torch
        self.n, self.stream = n, stream
        self.ev = [None] * (n + 1) if DRY else [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]

    def record(self, i):
        if DRY:
            self.ev[i] = time.perf_counter()
        else:
            self.ev[i].record(self.stream)

    def ms(self):
        if DRY:
            return [max((self.ev[i + 1] - self.ev[i]) * 1e3, 1e-6) for i in range(self.n)]
        return [self.ev[i].elapsed_time(self.ev[i + 1]) for i in range(self.n)]


class DryFunc:
    """Stand-in for the compiled evaluator in a dry run: touches nothing but the accumulator, to which every call adds the
    number of samples of the shard -- so the one collective's result is checkable (it must equal the job's sample count)."""

    def __init__(self, count):
        self.count = count

    def __call__(self, root, leaf):
        return root

    def accumulate(self, leaf, w, acc):
        acc += float(self.count)
        return acc

    def eval_tiled(self, root, leaf, n=None):
        return root

    def accumulate_tiled(self, leaf, w, acc, n=None):
        return self.accumulate(leaf, w, acc)

    def kernel_info(self):
        return {"last_kernel": "dry-run", "n_valu": [0, 0, 0]}

    def info(self):
        return {"max_live": 0, "spec_vgpr": 0, "spec_lds_bytes": 0, "spec_scratch_bytes": 0}


def probe_start(dev, seconds):
    """One sleeping wave on a side stream for about `seconds` (fdg_clock_probe_device): returns a handle for probe_stop, or None."""
    if DRY:
        return None
    try:
        import torch
        from feynmandiagram_jl_amd import capi
        side = torch.cuda.Stream(device=dev)
        ticks = torch.zeros(2, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        capi.clock_probe_device(min(30.0, max(2e-4, seconds)), ticks.data_ptr(), side.cuda_stream)
        return (side, ticks)
    except Exception:
        return None


def power_leg(step, seconds=1.5):
    """Socket power and shader clock (rocm-smi, from a side thread) while `step` runs back to back for about `seconds` -- AFTER and outside the timed
    region.  Returns {"power_w", "sclk_mhz", "power_cap_w", "samples"} or None (no rocm-smi, a different output format, a dry run): never an error."""
    if DRY:
        return None
    import json as _json
    import subprocess
    import threading
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    card = "card%d" % int(os.environ.get("LOCAL_RANK", "0"))
    got, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                d = _json.loads(subprocess.run([smi, "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
                c = d.get(card) or d[sorted(d)[0]]
                rec = {}
                for k, v in c.items():
                    kl = k.lower()
                    try:
                        if "max graphics package power" in kl: rec["cap"] = float(v)
                        elif "power (w)" in kl: rec["w"] = float(v)
                        elif "sclk clock speed" in kl: rec["sclk"] = float(str(v).strip("()").lower().replace("mhz", ""))
                    except ValueError:
                        pass
                if "w" in rec:
                    got.append(rec)
            except Exception:
                return
            time.sleep(0.05)
    try:
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                step()
            sync()
        stop[0] = True
        th.join(timeout=10)
        use = got[len(got) // 2:] if len(got) > 3 else (got[1:] if len(got) > 2 else got)          # (rocm-smi's power is a moving average: the later samples are the load's)
        if not use:
            return None
        out = {"power_w": sum(r["w"] for r in use) / len(use), "samples": len(use)}
        if all("sclk" in r for r in use): out["sclk_mhz"] = sum(r["sclk"] for r in use) / len(use)
        if "cap" in use[-1]: out["power_cap_w"] = use[-1]["cap"]
        return out
    except Exception:
        return None


def probe_stop(probe):
    """Shader clock in GHz the probe saw (s_memtime ticks per 100 MHz tick), None without a probe."""
    if not probe:
        return None
    probe[0].synchronize()
    c, w = (int(x) for x in probe[1].cpu())
    return c / w * 0.1 if w > 0 else None


class Case:
    """One workload resident on the device: handle, leaf batch (synthetic, Philox keyed by the global sample index), root buffer."""

    def __init__(self, workload, layout, B, dev, backend="isa", flags=0, sample_offset=0, placement="plain"):
        import torch
        import feynmandiagram_jl_amd as fd
        from feynmandiagram_jl_amd import capi, workloads
        self.workload, self.layout, self.B, self.dev = workload, layout, B, dev
        self.t = t = workloads.get(workload)
        self.st = t.stats()
        L, R = t.n_leaf, t.n_root
        self.sample_offset = sample_offset
        self.placement, self.pair, self.pair_error = placement, None, None
        if DRY:
            self.f, self.stream = DryFunc(B), None
            self.leaf = torch.zeros((1, L), dtype=torch.float64)
            self.root = torch.zeros((1, R, 64) if layout == "tile_major" else (min(B, 64), R), dtype=torch.float64)
            return
        self.f = fd.compile_table(t, specialize={"isa": "isa", "isa-autotune": "isa-autotune", "auto": "auto", "hip": True, "interp": False}[backend], flags=flags)
        self.stream = torch.cuda.current_stream()
        self.allocate()

    def allocate(self):
        """Allocates the leaf and root batches ONCE, as they come from the allocator, and fills the leaves (Philox keyed by the global
        sample index: the same values in every layout)."""
        import torch
        from feynmandiagram_jl_amd import capi
        B, L, R, dev = self.B, self.t.n_leaf, self.t.n_root, self.dev
        st = self.stream.cuda_stream
        if self.layout == "tile_major" and self.placement == "paired":
            # the library's own allocator for a tile-major batch (fdg_batch_alloc_pair): every window of the leaves gets a chunk of roots
            # behind which the handle's kernel was MEASURED at the fast rate; a plain allocation is the "@plain" row
            try:
                self.pair = self.f.tile_major_pair(B, dev, calibrate=True)
            except Exception as e:            # (a driver without the virtual-memory API, too little free memory: the line says so and uses a plain batch)
                self.pair, self.placement, self.pair_error = None, "plain", f"{type(e).__name__}: {e}"
                print(f"[bench] fdg_batch_alloc_pair failed ({self.pair_error}): plain allocation instead", file=sys.stderr)
            if self.pair is not None:
                self.leaf, self.root = self.pair.leaf, self.pair.root
                self.root.zero_()
                capi.fill_uniform_device_tiled(self.leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, self.sample_offset, st)
                return
        if self.layout == "sample_major" and self.placement == "paired":      # the same allocator for compile_Python's row-major [B, L] / [B, R]
            try:
                self.pair = self.f.row_major_pair(B, dev, calibrate=True)
            except Exception as e:
                self.pair, self.placement, self.pair_error = None, "plain", f"{type(e).__name__}: {e}"
            if self.pair is not None:
                self.leaf, self.root = self.pair.leaf, self.pair.root
                capi.fill_uniform_device(self.leaf.data_ptr(), B, L, self.leaf.stride(0), self.leaf.stride(1), 1234, self.sample_offset, st)
                return
        if self.layout == "leaf_major" and self.placement == "paired":        # ... and for a Julia column-major pair (one window: the whole batch)
            try:
                self.pair = self.f.leaf_major_pair(B, dev, calibrate=True)
            except Exception as e:
                self.pair, self.placement, self.pair_error = None, "plain", f"{type(e).__name__}: {e}"
            if self.pair is not None:
                self.leaf, self.root = self.pair.leaf, self.pair.root
                capi.fill_uniform_device(self.leaf.data_ptr(), B, L, self.leaf.stride(0), self.leaf.stride(1), 1234, self.sample_offset, st)
                return
        if self.layout == "tile_major":       # fdg_eval_device_tiled: [tile, value, sample in tile] -- a Julia Array{Float64,3}(64, L, cld(B, 64))
            T = (B + 63) // 64
            self.leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
            self.root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)      # (zeroed: lanes past B are never written)
            capi.fill_uniform_device_tiled(self.leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, self.sample_offset, st)
            return
        if self.layout == "sample_major":     # compile_Python's row-major [B, L] / [B, R]
            self.leaf = torch.empty((B, L), dtype=torch.float64, device=dev)
            self.root = torch.empty((B, R), dtype=torch.float64, device=dev)
        else:                                 # Julia column-major B x L / B x R matrices
            self.leaf = torch.empty((L, B), dtype=torch.float64, device=dev).t()
            self.root = torch.empty((R, B), dtype=torch.float64, device=dev).t()
        capi.fill_uniform_device(self.leaf.data_ptr(), B, L, self.leaf.stride(0), self.leaf.stride(1), 1234, self.sample_offset, st)

    def nbytes(self):
        return 0 if DRY else 8 * (((self.B + 63) // 64) * 64) * (self.t.n_leaf + self.t.n_root)

    def free(self):
        self.leaf = self.root = None
        if self.pair is not None:
            self.pair.free()
            self.pair = None

    def head(self, x, n):
        """The first n samples of a batch (leaf or root) as a host [n, C] array, whatever the layout."""
        import numpy as np
        if self.layout == "tile_major":
            T = (n + 63) // 64
            return np.ascontiguousarray(x[:T].permute(0, 2, 1).reshape(T * 64, x.shape[1])[:n].cpu().numpy())
        return np.ascontiguousarray(x[:n].cpu().numpy())

    def step(self):
        if self.layout == "tile_major":
            self.f.eval_tiled(self.root, self.leaf, self.B)
        else:
            self.f(self.root, self.leaf)

    def accumulate(self, w, acc):
        if self.layout == "tile_major":
            return self.f.accumulate_tiled(self.leaf, w, acc, self.B)
        return self.f.accumulate(self.leaf, w, acc)

    def settle(self, step=None, max_s=4.0):
        """Launches until the rate is steady: eight launches in a row within 1.5 % of each other (at most `max_s` seconds).  A batch is
        allocated and filled moments before it is measured, and the driver may still be wiping what the previous measurement released
        (settle_after_free waits by the clock; this waits by the kernel's own rate: profiles/r05_log_alloc_sensitivity_latency_bound.txt)."""
        if DRY:
            return
        import torch
        step = step or self.step
        t0 = time.perf_counter()
        hist = []
        while time.perf_counter() - t0 < max_s:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream); step(); e1.record(self.stream)
            e1.synchronize()
            hist.append(e0.elapsed_time(e1))
            if len(hist) >= 8 and min(hist[-8:]) > 0.985 * max(hist[-8:]):
                break

    def timed(self, steps, warm, step=None):
        """`warm` untimed launches, then `steps` launches bracketed by HIP events on the launch stream.  Returns ms per launch (list).
        Next to the timed launches one sleeping wave on a side stream (fdg_clock_probe_device) reads the shader clock the chip
        sustains under this load: `self.clock_ghz` (None when the probe could not run)."""
        import torch
        step = step or self.step
        for _ in range(warm):
            step()
        sync()
        self.clock_ghz = None
        probe = None
        if not DRY and steps >= 2:
            pre = Stamps(2, self.stream)          # two more untimed launches give the length of the region the probe has to cover
            pre.record(0); step(); pre.record(1); step(); pre.record(2)
            sync()
            # starting the probe costs a synchronisation and a few milliseconds of idle device: the memory side's clocks fall back and need tens of
            # milliseconds of load to return (round 5: rows measured right after that gap ran 5-10 % slow at a HIGH shader clock) -- so the
            # timed launches are preceded by ~40 ms of untimed ones inside the probe's window
            per = min(pre.ms())
            rewarm = min(400, max(10, int(40.0 / max(per, 1e-3)) + 1))
            probe = probe_start(self.dev, 0.9 * per * 1e-3 * (steps + rewarm))
            for _ in range(rewarm):
                step()
        ev = Stamps(steps, self.stream)
        ev.record(0)
        for i in range(steps):
            step()
            ev.record(i + 1)
        sync()
        self.clock_ghz = probe_stop(probe)
        return ev.ms()

    def parity_sample(self, n=2048):
        """The first n samples of the last launch against the oracle's restatement of the compiled evaluator (C, one thread)."""
        import numpy as np
        import oracle
        n = int(min(n, self.B))
        want = oracle.eval_static(self.t, self.head(self.leaf, n), np.zeros((n, self.t.n_root)))
        got = self.head(self.root, n)
        return bool(np.array_equal(got, want)), float(np.abs(got - want).max()) if n else 0.0, n


def observable_sum(root):
    """Column sums of the root matrix, the observable of the final reduction.  For a Julia-layout matrix (R long rows)
    torch's reduction runs one workgroup per row -- 50 ms for 4 x 10^8 doubles -- so the rows are summed in two stages."""
    if root.dim() == 3:                            # tile-major [T, R, 64]
        return root.sum(dim=2).sum(dim=0)
    if root.stride(0) != 1 or root.shape[0] < (1 << 16):
        return root.sum(dim=0)
    rt = root.t()                                  # [R, B], rows contiguous
    R, B = rt.shape
    c = 1 << 14
    nb = B // c
    acc = rt[:, :nb * c].reshape(R, nb, c).sum(dim=2).sum(dim=1)
    return acc + rt[:, nb * c:].sum(dim=1) if nb * c < B else acc


def roofline_of(st, B, avg_kernel_s, kernel, accumulate=False, ops_exec=None, clock_ghz=None):
    """Both roofs of one launch and the one that binds.  HBM: algorithmic bytes 8(L+R) per evaluation (8L when the
    roots are accumulated on chip) against 8 TB/s.  Vector ALU: the fold steps the kernel EXECUTES per evaluation
    (`ops_exec`, from fdg_graph_kernel_info: after value numbering, one v_add_f64 / v_mul_f64 each -- no FMA by
    contract) against 39.3e12 lane-op/s.  `bound` is the roof whose minimum time for this launch is larger; `frac`,
    `achieved`, `peak`, `unit` refer to it; the other roof's fraction is printed next to it."""
    bytes_per_eval = st["bytes_alg_accumulate"] if accumulate else st["bytes_alg"]
    gbs = bytes_per_eval * B / avg_kernel_s / 1e9
    frac_hbm = gbs / HBM_PEAK_GBS
    out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_hbm, "traffic": None,
           "kernel": kernel, "avg_kernel_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_eval * B,
           "frac_hbm": frac_hbm, "frac_valu": None, "ops_exec_per_eval": ops_exec}
    if ops_exec:
        tops = ops_exec * B / avg_kernel_s / 1e12
        out["frac_valu"] = tops / FP64_NOFMA_PEAK_TOPS
        out["valu_tops"] = tops
        out["frac_valu_of_measured_peak"] = tops / FP64_NOFMA_MEASURED_TOPS
        if out["frac_valu"] > frac_hbm:      # the launch cannot be shorter than ops / peak: the vector ALU is the binding roof
            out.update({"bound": "valu_fp64", "achieved": tops, "peak": FP64_NOFMA_PEAK_TOPS, "unit": "TFLOP/s", "frac": out["frac_valu"]})
    if ops_exec:
        out["power_roof_evals_per_s"] = (POWER_CAP_W - POWER_IDLE_W) / (E_BYTE_J * bytes_per_eval + E_OP_J * ops_exec)
        out["frac_power"] = (B / avg_kernel_s) / out["power_roof_evals_per_s"]
    if clock_ghz:
        # the shader clock the chip sustained during the timed launches (one sleeping wave on a side stream): the graphs at the
        # compute/memory ridge run against the power budget (1.8-1.9 GHz), and the vector-ALU roof scales with it
        out["clock_ghz"] = clock_ghz
        if out.get("frac_valu") is not None:
            out["frac_valu_at_clock"] = out["valu_tops"] / (FP64_NOFMA_PEAK_TOPS * clock_ghz / SPEC_CLOCK_GHZ)
    return out


def kernel_of(f, slot=0):
    """(name of the kernel the last call launched, fold steps it executes per evaluation) from the handle itself."""
    try:
        ki = f.kernel_info()
    except Exception:
        return "", None
    name = ki["last_kernel"]
    if name == "fdg_isa_eval_pool":       # the pooled cooperative variant: fold steps of all its waves together
        return name, (ki.get("pool_valu") or None)
    if name == "fdg_isa_eval_rl":         # the linear row-major variant has its own program
        return name, (ki.get("rl_valu") or None)
    slot = 1 if "_acc" in name else 2 if name.endswith("_rm") else 0
    return name, (ki["n_valu"][slot] or None)


TRAFFIC_FILE = "r06_traffic.json"       # this round's PMC summaries only: a workload that is not in it gets traffic = null, never an older round's figure


def attach_traffic(roof, workload, layout, B, avg_kernel_s):
    """HBM bytes per launch from the rocprofv3 --pmc passes of the same command (bench.py cannot collect counters on
    itself): profiles/r04_traffic.json holds bytes per evaluation, scaled here to this batch -- a value from the named
    profile, not a measurement of this run."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE))).get(workload if layout == "leaf_major" else workload + ":" + layout)
    except (OSError, ValueError):
        return
    if tr and tr.get("layout", "leaf_major") == layout:
        roof["traffic"] = tr["bytes_per_eval"] * B
        roof["traffic_gbs"] = tr["bytes_per_eval"] * B / avg_kernel_s / 1e9
        roof["traffic_frac"] = roof["traffic_gbs"] / HBM_PEAK_GBS
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
        roof["traffic_source"] = "profiles/" + TRAFFIC_FILE + " (separate --pmc pass, not this run)"
        roof["traffic_source_detail"] = ("(2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (" + tr.get("source", "") +
                                         "), per evaluation, scaled to this batch")


def measured_read(dev):
    """The memory system's ceiling for a READ stream (the evaluator's traffic is 95 % reads): 4 GiB through fdg_read_device (8 bytes per lane, non-temporal)."""
    import torch
    from feynmandiagram_jl_amd import capi
    a = torch.zeros(1 << 29, dtype=torch.float64, device=dev)
    sink = torch.zeros(1, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        capi.read_device(a.data_ptr(), a.numel(), sink.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        capi.read_device(a.data_ptr(), a.numel(), sink.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    return 20 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9


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
    roofline fraction, and a bitwise check of a sample against the oracle.  A layout ending in "+fma" is the SEPARATELY DISCLOSED
    contracted mode (FDG_SPEC_FAST_MATH: a product used once by a sum becomes v_fma_f64): not bit-identical -- the row carries the
    measured max |d| / S_k (S_k: the root's sum of absolute terms, BASELINE.json's 1e-12 scale) instead of a bitwise verdict."""
    import torch
    try:
        from feynmandiagram_jl_amd import capi
        fma = layout.endswith("+fma")
        plain = layout.endswith("@plain")        # the headline's workload at the headline's size on a PLAIN allocation (what hipMalloc hands out)
        # "@1e8" / "@1e8*" (round 6): the reference's own layouts -- a Julia column-major B x L Matrix, compile_Python's row-major
        # [B, L] -- at the HEADLINE's batch size, on a plain allocation / through fdg_batch_alloc_pair
        full_plain, full_paired = layout.endswith("@1e8"), layout.endswith("@1e8*")
        lay = layout[:-4] if (fma or full_plain) else (layout[:-6] if plain else (layout[:-5] if full_paired else layout))
        # the memory-bound graphs with root stores get their batch from the library's allocator (fdg_batch_alloc_pair), as the headline does
        paired = (((workload, lay) in PAIRED_ROWS or (PAIR_ALL and lay in ("tile_major", "sample_major", "leaf_major"))) and not plain and not fma and not full_plain) or full_paired
        big = plain or full_plain or full_paired
        c = Case(workload, lay, DEFAULT_B[workload] if big else (16_000_000 if workload == "parquet_sigma4" else DEFAULT_B.get(workload, 1_000_000)), dev,
                 flags=capi.FDG_SPEC_FAST_MATH if fma else 0, placement="paired" if paired else "plain")
        settle_after_free(c.nbytes())        # (by the clock: whatever was released to make room for this batch is being wiped)
        c.settle()                           # (by the kernel's own rate)
        ms = c.timed(steps, warm)
        avg = sum(ms) / len(ms) / 1e3
        ok, dev_max, n = c.parity_sample()
        dev_over_sk = None
        if fma:
            import numpy as np
            import oracle
            h_leaf = c.head(c.leaf, n)
            want = oracle.eval_static(c.t, h_leaf)
            dev_over_sk = float(np.max(np.abs(c.head(c.root, n) - want) / np.maximum(1.0, oracle.root_scale(c.t, h_leaf))))
        kern, ops_exec = kernel_of(c.f)
        # socket power and shader clock from rocm-smi while the row's launch runs back to back (after its timed launches).  The sleeping-wave probe
        # reads the clock of the ONE XCD it landed on and disagreed with rocm-smi on the ridge rows (gv_sigma5 tile-major 2.40 GHz against
        # 1655 MHz): rocm-smi's figure is the row's clock_ghz, the probe's stays in the detail file as clock_ghz_probe
        pw = power_leg(c.step, seconds=POWER_LEG_S) if POWER_LEG_S > 0 else None
        smi_ghz = pw["sclk_mhz"] / 1e3 if pw and pw.get("sclk_mhz") else None
        roof = roofline_of(c.st, c.B, avg, kern, ops_exec=ops_exec, clock_ghz=smi_ghz or c.clock_ghz)
        roof["clock_source"] = "rocm-smi sclk" if smi_ghz else ("sleeping-wave probe" if c.clock_ghz else None)
        if c.clock_ghz:
            roof["clock_ghz_probe"] = c.clock_ghz
        if pw:
            roof.update({"power_w": pw["power_w"], "sclk_mhz": pw.get("sclk_mhz"), "power_cap_w": pw.get("power_cap_w"), "power_samples": pw["samples"]})
        attach_traffic(roof, workload, lay, c.B, avg)
        if copy_gbs:
            roof["frac_of_measured_copy"] = roof["achieved"] / copy_gbs
        info = c.f.info()
        out = {"workload": workload + WORKLOAD_NOTES.get(workload, ""), "layout": layout, "value": c.B / avg, "unit": "evals/s",
               "samples_per_launch": c.B, "timed_launches": steps, "warmup_launches": warm, "avg_kernel_ms": avg * 1e3,
               "n_leaf": c.t.n_leaf, "n_node": c.t.n_node, "n_root": c.t.n_root, "flops_per_eval": c.st["flops_alg"], "bytes_per_eval": c.st["bytes_alg"],
               "roofline": roof,
               "valu_fp64_tflops": c.st["flops_alg"] * c.B / avg / 1e12,
               "kernel_info": {k: info[k] for k in ("max_live", "spec_vgpr", "spec_lds_bytes")},
               "gpu_matches_cpu_bitwise": ok, "max_abs_dev": dev_max, "parity_samples": n,
               "placement": ("fdg_batch_alloc_pair" if c.pair is not None else "plain allocation")}
        if c.pair is not None:
            out["placement_info"] = {k: c.pair.info[k] for k in ("n_chunk", "n_candidate", "n_probe", "n_matched", "calibrated", "seconds", "seconds_settling")}
        if fma:
            out["contracted"] = True
            out["max_dev_over_Sk"] = dev_over_sk
            out["gpu_matches_cpu_bitwise"] = None if not ok else True      # (not a claim of this mode; BASELINE's bar is 1e-12 of S_k)
        if full_plain and lay in ("leaf_major", "sample_major"):
            # what tile_major! costs (fdg_repack_tile_major: one pass at copy speed) and what the same batch then runs at tile-major, on a plain allocation
            try:
                import feynmandiagram_jl_amd as fd
                T = (c.B + 63) // 64
                tl = torch.empty((T, c.t.n_leaf, 64), dtype=torch.float64, device=dev)
                tr = torch.zeros((T, c.t.n_root, 64), dtype=torch.float64, device=dev)
                for _ in range(2):
                    fd.GraphFunc.tile_major_(tl, c.leaf)
                sync()
                rp = Stamps(5, c.stream)
                rp.record(0)
                for i in range(5):
                    fd.GraphFunc.tile_major_(tl, c.leaf)
                    rp.record(i + 1)
                sync()
                rp_ms = sum(rp.ms()) / 5
                for _ in range(30):
                    c.f.eval_tiled(tr, tl, c.B)
                sync()
                ev2 = Stamps(steps, c.stream)
                ev2.record(0)
                for i in range(steps):
                    c.f.eval_tiled(tr, tl, c.B)
                    ev2.record(i + 1)
                sync()
                ev_ms = sum(ev2.ms()) / steps
                back = torch.empty_strided(c.root.size(), c.root.stride(), dtype=torch.float64, device=dev)
                fd.GraphFunc.from_tile_major_(back, tr)
                same = bool(torch.equal(back, c.root))
                out["repack"] = {"what": "fdg_repack_tile_major of this batch's leaves (GraphFunc.tile_major_ / Julia tile_major!), then fdg_eval_device_tiled on the result (plain allocation), roots back through fdg_unpack_tile_major",
                                 "repack_ms": rp_ms, "repack_gbs_read_plus_written": 2 * 8 * c.B * c.t.n_leaf / (rp_ms * 1e-3) / 1e9,
                                 "eval_tiled_ms": ev_ms, "eval_tiled_frac_hbm": c.st["bytes_alg"] * c.B / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "in_place_ms": avg * 1e3, "evaluations_to_amortise": (rp_ms / max(avg * 1e3 - ev_ms, 1e-9)) if avg * 1e3 > ev_ms else None,
                                 "roots_equal_in_place_bits": same}
                del tl, tr, back
            except Exception as e:
                out["repack"] = {"error": f"{type(e).__name__}: {e}"}
        freed = c.nbytes()
        c.free()
        del c
        torch.cuda.empty_cache()
        settle_after_free(freed)
        return out
    except Exception as e:                      # secondary: never takes the headline line down
        return {"workload": workload, "layout": layout, "error": f"{type(e).__name__}: {e}"}


def config5_steps(world, per_gpu=None, total=CONFIG5_TOTAL_SAMPLES):
    """Steps per rank so that all ranks together evaluate BASELINE.json's 10^9 samples (independent of --steps)."""
    per_gpu = per_gpu or DEFAULT_B["gv_sigma5"]
    return max(1, -(-total // (per_gpu * world)))


def config5(dev, rank, world, dist, comm, steps, warm):
    """BASELINE.json config 5 (example/benchmark_GV.jl as BASELINE.json words it: the GV 5th-order self-energy, 10^9 samples
    sharded over the GPUs, one final reduce): per step fdg_accumulate_device on this rank's shard -- weighted
    accumulation inside the evaluator, roots never reach HBM --, after the last step ONE all-reduce of R doubles.
    `steps` = config5_steps(world): 500 steps of 2e6 samples on one GPU, 63 on eight."""
    import torch
    from feynmandiagram_jl_amd.sharding import reduce_observable, shard_range
    try:
        B = DEFAULT_B["gv_sigma5"]
        start, count = shard_range(B * world, rank, world)
        c = Case("gv_sigma5", "tile_major", count, dev, sample_offset=start)
        w = torch.rand(1 if DRY else count, dtype=torch.float64, device=dev)
        acc = torch.zeros(c.t.n_root, dtype=torch.float64, device=dev)
        settle_after_free(c.nbytes())
        c.settle(lambda: c.accumulate(w, acc))
        pre = Stamps(warm, c.stream)
        pre.record(0)
        for i in range(warm):
            c.accumulate(w, acc)
            pre.record(i + 1)
        acc.zero_()
        sync()
        if dist:
            dist.barrier()
            sync()
        per = min(pre.ms()) if warm else 1.0
        rewarm = min(400, max(10, int(40.0 / max(per, 1e-3)) + 1)) if not DRY else 0
        probe = probe_start(dev, 0.9 * per * 1e-3 * (steps + rewarm)) if warm else None     # (a sleeping wave on a side stream: the clock under this load)
        for _ in range(rewarm):               # (the synchronisations above left the device idle for a moment: see Case.timed)
            c.accumulate(w, acc)
        acc.zero_()
        sync()                                # (the wall clock below must not start while these are still running; a bare synchronisation is microseconds)
        ev = Stamps(steps, c.stream)
        t0 = time.perf_counter()
        ev.record(0)
        for i in range(steps):
            c.accumulate(w, acc)
            ev.record(i + 1)
        reduce_observable(acc, comm=comm)     # the one collective: R doubles over xGMI (RCCL)
        sync()
        if dist:
            dist.barrier()
            sync()
        elapsed = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        ms = ev.ms()
        avg = sum(ms) / len(ms) / 1e3
        total = float(count) * steps * world
        shards = [[int(start), int(count)]]
        if dist:                                  # (after the timed region) every rank's contiguous range of a step's samples, for the record
            sh = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(sh, torch.tensor([start, count], dtype=torch.int64, device=dev))
            shards = [[int(x[0]), int(x[1])] for x in sh]
        observable = [float(x) for x in acc.cpu()]
        kern, ops_exec = kernel_of(c.f)
        probe_ghz = probe_stop(probe)
        pw = power_leg(lambda: c.accumulate(w, acc), seconds=POWER_LEG_S) if (POWER_LEG_S > 0 and world == 1) else None      # (after the timed region; acc is dead by then)
        smi_ghz = pw["sclk_mhz"] / 1e3 if pw and pw.get("sclk_mhz") else None
        roof = roofline_of(c.st, count, avg, kern + " + fdg_reduce_lane_partials", accumulate=True, ops_exec=ops_exec, clock_ghz=smi_ghz or probe_ghz)
        roof["clock_source"] = "rocm-smi sclk" if smi_ghz else ("sleeping-wave probe" if probe_ghz else None)
        if probe_ghz:
            roof["clock_ghz_probe"] = probe_ghz
        if pw:
            roof.update({"power_w": pw["power_w"], "sclk_mhz": pw.get("sclk_mhz"), "power_cap_w": pw.get("power_cap_w"), "power_samples": pw["samples"]})
        out = {"workload": "gv_sigma5" + WORKLOAD_NOTES["gv_sigma5"], "value": total / elapsed, "unit": "samples/s (whole job)", "n_gpus": world,
               "steps": steps, "warmup": warm, "samples_per_step_per_gpu": count, "total_samples": total,
               "shard_offset_rank0": start, "shards_of_a_step": shards, "baseline_total_samples": CONFIG5_TOTAL_SAMPLES,
               "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "roofline_rank0": roof,
               "observable": observable,
               "layout": "tile_major", "what": "fdg_accumulate_device_tiled per step on the rank's shard (tile-major batch); one all-reduce of R doubles after the last step, inside the timed region"}
        freed = c.nbytes()
        c.free()
        del c, w
        if not DRY:
            torch.cuda.empty_cache()
            settle_after_free(freed)
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
    ap.add_argument("--layout", default="tile_major", choices=["tile_major", "sample_major", "leaf_major"],
                    help="tile_major = the batch as [tile of 64 samples][leaf][sample in tile] (a Julia Array{Float64,3}(64, L, cld(B, 64)); "
                         "fdg_eval_device_tiled): what a Monte-Carlo driver that owns its batch allocates, and the default; "
                         "leaf_major = a Julia column-major B x L matrix; sample_major = compile_Python's row-major [B, L]")
    ap.add_argument("--placement", default="paired", choices=["paired", "plain"],
                    help="headline batch (tile-major only): paired = fdg_batch_alloc_pair, the library's allocator that times (leaf window, root chunk) "
                         "pairs and maps a fast root chunk behind every window; plain = torch.empty, whatever hipMalloc hands out")
    ap.add_argument("--pair-all", action="store_true", help="experiment: every tile-major / row-major secondary row allocates through fdg_batch_alloc_pair")
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
    ap.add_argument("--secondary", default="", help="comma-separated workload:layout pairs to measure after the headline instead of the full list")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--power-seconds", type=float, default=1.5, help="per secondary row: seconds of back-to-back launches sampled with rocm-smi (socket power, sclk); 0 = off")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device work: one process per rank on the CPU (gloo), the evaluator replaced by a stub that adds the shard's "
                         "sample count to the accumulator; checks sharding, the one collective and the stdout line (tests/test_bench_line.py)")
    args = ap.parse_args()
    global DRY, PAIR_ALL, POWER_LEG_S
    DRY = bool(args.dry_run)
    PAIR_ALL = bool(args.pair_all)
    POWER_LEG_S = float(args.power_seconds)

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
        dist.init_process_group("gloo" if DRY else "nccl", rank=rank, world_size=world)   # nccl == RCCL on ROCm
    if DRY:
        dev = torch.device("cpu")
        args.no_cpu_baseline = args.no_mc_step = True
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    comm = make_comm(rank, world) if (dist and args.comm == "fdg" and not DRY) else None

    if args.interp:
        args.backend = "interp"
    # Samples resident per step.  Decimal sizes on purpose: BASELINE.json's sample counts are decimal (25 steps of the
    # default = config 3's 10^8 samples), and a leaf-major matrix whose column stride is a power of two aliases HBM
    # channels (measured: -3 % on the default workload, -14 % on sigma2).
    B = args.samples or DEFAULT_B.get(args.workload, 1_000_000)
    t_probe = workloads.get(args.workload)
    free_b = (1 << 62) if DRY else torch.cuda.mem_get_info(dev)[0]
    while not args.samples and 8 * B * (t_probe.n_leaf + t_probe.n_root) > 0.6 * free_b and B > 1_000_000:
        B //= 2                         # (a 288 GB device holds config 3's 1e8 samples of 84 leaves, 70 GB, with room to spare)
    # per-rank Philox offset: results do not depend on how samples are sharded
    start, count = shard_range(B * world, rank, world)          # weak scaling: B samples per GPU
    assert count == B
    paired = args.layout == "tile_major" and args.placement == "paired" and args.backend in ("isa", "isa-autotune") and not DRY
    case = Case(args.workload, args.layout, B, dev, backend=args.backend, flags=capi.FDG_SPEC_FAST_MATH if args.fast_math else 0,
                sample_offset=start, placement="paired" if paired else "plain")
    if args.backend == "isa-autotune":
        args.backend = "isa"
    if args.layout == "tile_major" and args.backend not in ("isa", "auto") and not DRY:
        raise SystemExit("--layout tile_major needs the ISA back end (fdg_eval_device_tiled)")
    t, st, f, leaf, root, stream = case.t, case.st, case.f, case.leaf, case.root, case.stream
    L, R = t.n_leaf, t.n_root
    step = case.step

    step()
    _ = observable_sum(root)              # load the reduction used for the final observable now: a pause between the
    sync()                                # warm-up and the timed steps would let the clocks fall back
    # Clock settling: after idle the first ~50 launches run at transient clocks (boost, then throttle, then the
    # sustained state: 1.10 -> 1.40 -> 1.05-1.15 ms per launch on the default workload).  The timed steps are meant to
    # show the sustained rate, so at least 60 untimed launches precede them whatever --warmup says (disclosed in the
    # JSON line as config.settle_steps; they are the same step as the warm-up and the timed ones).
    settle = max(0, 60 - args.warmup)
    for _ in range(settle + args.warmup):
        step()
    sync()
    if dist:
        dist.barrier()
        sync()
    ev = Stamps(args.steps, stream)
    t0 = time.perf_counter()
    ev.record(0)
    for i in range(args.steps):
        step()
        ev.record(i + 1)                  # same stream the kernel is launched on
    acc = observable_sum(root)            # final observable accumulation
    reduce_observable(acc, comm=comm)     # the one collective: R doubles over xGMI (RCCL)
    sync()
    if dist:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = ev.ms()
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
        "data": "synthetic" if not DRY else "dry-run: no device work, timings meaningless",
        "config": {"workload": args.workload + WORKLOAD_NOTES.get(args.workload, ""),
                   "graph": t.name, "n_leaf": L, "n_node": t.n_node, "n_edge": t.n_edge, "n_root": R,
                   "flops_per_eval": st["flops_alg"], "bytes_per_eval": st["bytes_alg"],
                   "samples_per_step_per_gpu": B, "layout": args.layout, "settle_steps": settle, "shard_offset_rank0": start,
                   "kernel": {"isa": "fdg_isa_eval (per-graph gfx950 assembly)", "hip": "fdg_spec (per-graph HIP source, hiprtc)",
                              "auto": "fdg_isa_eval, or its HIP-source companion fdg_spec_sm for row-major input of small graphs",
                              "interp": "fdg_interp (table interpreter)"}[args.backend],
                   "parallelism": f"samples sharded x{world}, one all-reduce of {R} doubles",
                   "parity": PARITY_NOTE},
    }
    if not DRY:
        try:                              # which physical device this line was measured on (the round's lines come from several boxes)
            out["config"]["device"] = str(torch.cuda.get_device_properties(dev).uuid)[-12:]
        except Exception:
            pass
    kname, ops_exec = kernel_of(f)        # the kernel the library actually launched, and what it executes per evaluation
    copy_gbs = None
    if rank == 0:
        out["roofline"] = roofline_of(st, B, avg_kernel_s, kname, ops_exec=ops_exec)
        achieved = st["bytes_alg"] * B / avg_kernel_s / 1e9
        fr = sorted(st["bytes_alg"] * B / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in kern_ms)
        # the same fraction from the driver-facing wall clock of the timed region (ms_per_step: the K launches + the final observable sum + the one reduce)
        out["roofline"]["frac_from_ms_per_step"] = st["bytes_alg"] * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS
        out["roofline"]["frac_hbm_min_over_steps"] = fr[0]
        out["roofline"]["frac_hbm_max_over_steps"] = fr[-1]
        out["roofline"]["frac_hbm_median_over_steps"] = fr[len(fr) // 2]
        out["roofline"]["frac_hbm_p05_over_steps"] = fr[len(fr) // 20]      # (a launch in a hundred runs 10 % slow now and then: the minimum is that launch)
        out["roofline"]["frac_hbm_of_each_step"] = [round(st["bytes_alg"] * B / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for m in kern_ms]     # (detail file only)
        if case.pair is not None:
            pi = case.pair.info
            out["roofline"]["placement"] = ("fdg_batch_alloc_pair: leaves one hipMalloc, roots mapped in %d chunks chosen by timing this kernel on (leaf window, root chunk) pairs; "
                                            "one batch, allocated once" % pi["n_chunk"])
            out["roofline"]["placement_info"] = {"windows": pi["n_chunk"], "tiles_per_window": pi["chunk_tiles"], "root_chunks_drawn": pi["n_candidate"], "fillers": pi["n_filler"],
                                                 "pairs_timed": pi["n_probe"], "windows_at_fast_level": pi["n_matched"], "contrast_found": bool(pi["calibrated"]),
                                                 "pair_frac_best": pi["gbs_fast"] / HBM_PEAK_GBS, "pair_frac_worst": pi["gbs_slow"] / HBM_PEAK_GBS,
                                                 "draw_order_pairs_frac_mean": pi["gbs_before_mean"] / HBM_PEAK_GBS, "draw_order_pairs_frac_min": pi["gbs_before_min"] / HBM_PEAK_GBS,
                                                 "mapped_pairs_frac_mean": pi["gbs_after_mean"] / HBM_PEAK_GBS, "mapped_pairs_frac_min": pi["gbs_after_min"] / HBM_PEAK_GBS,
                                                 "seconds": pi["seconds"], "seconds_settling": pi["seconds_settling"]}
        else:
            out["roofline"]["placement"] = "single allocation, as it came from the allocator (no trials)" + (f"; fdg_batch_alloc_pair failed: {case.pair_error}" if case.pair_error else "")
        try:
            if DRY:
                raise RuntimeError("dry run")
            copy_gbs = measured_copy(dev)
            out["roofline"]["measured_copy_gbs"] = copy_gbs
            out["roofline"]["measured_copy_kernel"] = "fdg_copy_device (16 B per lane, non-temporal loads and stores, 2 GiB, read + write counted)"
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            read_gbs = measured_read(dev)
            out["roofline"]["measured_read_gbs"] = read_gbs            # (a read-only non-temporal stream: what the memory system gives the evaluator's kind of traffic)
            out["roofline"]["frac_of_measured_read"] = achieved / read_gbs
        except RuntimeError:
            pass
        if not DRY:
            # the clock the chip sustains under the headline launch: ten more launches AFTER the contract's timed region, with the
            # probe wave on a side stream (the timed region itself runs without it)
            case.timed(min(10, max(2, args.steps)), 0)
            if case.clock_ghz:
                out["roofline"]["clock_ghz"] = case.clock_ghz
                if out["roofline"].get("frac_valu") is not None:
                    out["roofline"]["frac_valu_at_clock"] = out["roofline"]["valu_tops"] / (FP64_NOFMA_PEAK_TOPS * case.clock_ghz / SPEC_CLOCK_GHZ)
        if not DRY and world == 1:
            # what the package draws under the headline launch (rocm-smi; after and outside the timed region): the rows of this line run at the cap
            pw = power_leg(case.step)
            if pw:
                out["roofline"].update({"power_w": pw["power_w"], "power_cap_w": pw.get("power_cap_w"), "sclk_mhz": pw.get("sclk_mhz"), "power_samples": pw["samples"]})
        if args.backend == "isa":
            attach_traffic(out["roofline"], args.workload, args.layout, B, avg_kernel_s)
        out["valu_fp64"] = {"achieved_tflops": st["flops_alg"] * B / avg_kernel_s / 1e12,
                            "peak_tflops_fma": FP64_VALU_PEAK_TFLOPS,
                            "note": "secondary ceiling: add/mul only (no FMA contraction allowed), so the usable peak is half"}
        out["kernel_info"] = {k: info[k] for k in ("max_live", "spec_vgpr", "spec_lds_bytes", "spec_scratch_bytes")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(t, case, args.cpu_seconds)
        if world == 1 and args.backend == "isa" and not DRY and not args.fast_math:
            # The Monte-Carlo use of the same launch (after and outside the timed region, never part of `value`): weighted
            # accumulation inside the evaluator on the SAME resident batch -- fdg_accumulate_device: roots never reach HBM, so
            # the read stream has no stores in it.  Algorithmic bytes per sample: 8 L + 8 (the weight).
            try:
                wgt = torch.rand(B, dtype=torch.float64, device=dev)
                accv = torch.zeros(R, dtype=torch.float64, device=dev)
                n_acc = int(max(3, min(args.steps, 30)))
                for _ in range(5):
                    case.accumulate(wgt, accv)
                sync()
                eva = Stamps(n_acc, stream)
                eva.record(0)
                for i in range(n_acc):
                    case.accumulate(wgt, accv)
                    eva.record(i + 1)
                sync()
                ms_acc = eva.ms()
                avg_acc = sum(ms_acc) / len(ms_acc) / 1e3
                ka, ops_a = kernel_of(f)
                ra = roofline_of({"bytes_alg": 8 * (L + 1), "bytes_alg_accumulate": 8 * (L + 1)}, B, avg_acc, ka, accumulate=True, ops_exec=ops_a)
                out["accumulate"] = {"value": B / avg_acc, "unit": "samples/s", "samples_per_launch": B, "timed_launches": n_acc, "avg_kernel_ms": avg_acc * 1e3,
                                     "roofline": ra, "what": "fdg_accumulate_device on the headline's resident batch (acc[k] += sum_b w_b root_k(b), roots never written)"}
                del wgt, accv
            except Exception as e:
                out["accumulate"] = {"error": f"{type(e).__name__}: {e}"}
    freed = case.nbytes() + (8 * B if not DRY else 0)
    case.free()
    del case, leaf, root, f, step
    if not DRY:
        torch.cuda.empty_cache()
        settle_after_free(freed)
    # ---- after and outside the headline's timed region ---------------------------------------------------------
    if args.backend == "isa" and not args.no_secondary and not args.fast_math:
        # config 5 runs on every rank (its one collective needs them all); the rest on rank 0 at N = 1 only
        c5 = config5(dev, rank, world, dist if world > 1 or os.environ.get("FDG_BENCH_FORCE_DIST") else None, comm, steps=config5_steps(world), warm=30)
        if rank == 0:
            out["config5"] = c5
        if rank == 0 and world == 1 and not DRY:
            sec = []
            head = (args.workload, args.layout)
            full = ((args.workload, "tile_major@plain"), ("parquet_sigma4", "leaf_major@1e8"), ("parquet_sigma4", "sample_major@1e8"), ("parquet_sigma4", "sample_major@1e8*"),
                    ("parquet_sigma4", "tile_major"), ("parquet_sigma4", "leaf_major"), ("parquet_sigma4", "sample_major"), ("parquet_sigma4_dyn", "tile_major"),
                    ("parquet_sigma4_insdyn", "tile_major"), ("parquet_sigma4_taylor2", "tile_major"), ("parquet_sigma4_taylor2", "tile_major+fma"),
                    ("parquet_sigma5", "tile_major"), ("parquet_sigma5", "sample_major"), ("parquet_ver4_4", "tile_major"), ("gv_ver4_4", "tile_major"), ("gv_ver4_4", "leaf_major"), ("sigma2", "tile_major"), ("sigma4_standin", "leaf_major"), ("gv_sigma4", "tile_major"),
                    ("gv_sigma5", "tile_major"), ("gv_sigma5", "leaf_major"), ("gv_sigma5", "tile_major+fma"),
                    ("gv_sigma6", "leaf_major"), ("gv_sigma4_taylor2", "tile_major"), ("gv_sigma4_taylor2", "sample_major"))
            for wl, lay in (tuple(tuple(x.split(":")) for x in args.secondary.split(",")) if args.secondary else full):
                # (the headline's own workload is measured once more as a secondary row when the headline ran another batch size: the
                #  row-major row of parquet_sigma4 then has its leaf-major partner at the same 1.6e7 samples, in the same process)
                if lay.endswith("@plain") and not (paired and wl in DEFAULT_B and 8 * DEFAULT_B[wl] * (t.n_leaf + t.n_root) < 0.45 * torch.cuda.mem_get_info(dev)[0]):
                    continue            # (the plain-allocation partner of a paired headline only)
                if "@1e8" in lay and not (wl in DEFAULT_B and 8 * DEFAULT_B[wl] * (t.n_leaf + t.n_root) < 0.45 * torch.cuda.mem_get_info(dev)[0]):
                    continue            # (the headline-sized rows need the headline's memory)
                if (wl, lay) != head or (not args.secondary and wl == "parquet_sigma4" and B != 16_000_000):
                    sec.append(secondary_case(wl, lay, dev, copy_gbs=copy_gbs))
            out["secondary"] = sec
            out["secondary_note"] = ("layout with a * = batch from fdg_batch_alloc_pair like the headline's (the memory-bound graphs with root stores), otherwise a plain allocation; "
                                     "measured in this process after the headline's timed region (30 untimed + 20 timed launches each, HIP events on the "
                                     "launch stream); never part of `value`.  " + PARITY_NOTE)
    if rank == 0:
        if world == 1 and not args.no_mc_step and args.backend == "isa":
            out["mc_step"] = mc_step(t, args.workload, min(B, 16_000_000), dev, args.fast_math)
        # the full objects go to bench_detail.json (and stderr); stdout gets ONE short line the driver can parse
        try:
            with open(DETAIL_PATH, "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError as e:
            print(f"[bench] cannot write {DETAIL_PATH}: {e}", file=sys.stderr)
        print("[bench] detail: " + json.dumps(out), file=sys.stderr)
        os.write(real_stdout, (compact_line(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()


def _r(x, n=4):
    """Short number for the stdout line: n significant digits."""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float(f"{x:.{n}g}")


def compact_line(full):
    """The ONE stdout line: the contract's keys, the headline's roofline and cpu_baseline, and the other measurements as
    short rows -- everything else stays in bench_detail.json.  Kept under LINE_LIMIT bytes (tests/test_bench_line.py)."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data") if k in full}
    line["value"] = _r(line.get("value"), 6)
    line["ms_per_step"] = _r(line.get("ms_per_step"), 6)
    cfg = full.get("config", {})
    line["config"] = {"workload": SHORT_NOTES.get(cfg.get("workload", "").split(" ")[0], cfg.get("workload", "").split(" ")[0])[:120]}
    for k in ("n_leaf", "n_node", "n_root", "samples_per_step_per_gpu", "layout", "settle_steps", "parallelism", "device"):
        if k in cfg:
            line["config"][k] = cfg[k]
    roof = full.get("roofline")
    if roof:
        keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source", "kernel", "avg_kernel_ms",
                "frac_hbm", "frac_valu", "frac_from_ms_per_step", "frac_hbm_min_over_steps", "measured_read_gbs", "frac_of_measured_read",
                "ops_exec_per_eval", "frac_power", "power_w", "power_cap_w", "sclk_mhz")       # (measured_copy_gbs, the probe's clock_ghz, frac_valu_at_clock: detail file)
        line["roofline"] = {k: (_r(roof[k], 5) if k != "traffic_source" else str(roof[k])[:26]) for k in keep if k in roof}
        if roof.get("placement"):
            pi = roof.get("placement_info")
            line["roofline"]["placement"] = ("fdg_batch_alloc_pair, one batch: %d/%d windows fast (%.3f)" % (pi["windows_at_fast_level"], pi["windows"], pi["pair_frac_best"])
                                             if pi else "single allocation")
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 5), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample", ""))[:64], "gpu_matches_cpu_bitwise": cb.get("gpu_matches_cpu_bitwise")}
    sec = full.get("secondary")
    if sec:
        # (sclk_ghz / power_w: rocm-smi next to the row's own launches, round 6; the self-calibrated frac_power of round 5 stays in bench_detail.json as a diagnostic)
        line["secondary_cols"] = ["workload", "layout", "Mevals_per_s", "bound", "frac", "frac_hbm", "frac_valu", "traffic_ratio", "bitwise", "sclk_ghz", "power_w"]
        rows = []
        for e in sec:
            if "error" in e:
                rows.append([e.get("workload", "?")[:24], e.get("layout"), None, "error", None, None, None, None, False, None, None])
                continue
            r = e["roofline"]
            lay_short = e["layout"].rstrip("*")
            for long_, short_ in (("leaf_major", "lm"), ("sample_major", "rm"), ("tile_major", "tm")):
                lay_short = lay_short.replace(long_, short_)
            rows.append([e["workload"].split(" ")[0], lay_short + ("*" if e.get("placement") == "fdg_batch_alloc_pair" else ""), _r(e["value"] / 1e6),
                         {"hbm": "hbm", "valu_fp64": "valu"}.get(r["bound"], r["bound"]), _r(r["frac"], 3), _r(r.get("frac_hbm"), 3),
                         _r(r.get("frac_valu"), 3), _r(r.get("traffic_over_algorithmic"), 3),
                         ({True: 1, False: 0}.get(e.get("gpu_matches_cpu_bitwise"), e.get("gpu_matches_cpu_bitwise")) if not e.get("contracted") else "fma:%.0e" % (e.get("max_dev_over_Sk") or 0.0)),
                         _r(r.get("clock_ghz"), 3), (int(round(r["power_w"])) if r.get("power_w") else None)])
        line["secondary"] = rows
    rp = {}
    for e in (sec or []):
        r = e.get("repack") if isinstance(e, dict) else None
        if r and "error" not in r:
            rp["lm" if e["layout"].startswith("leaf_major") else "rm"] = [_r(r["repack_ms"], 3), _r(r["eval_tiled_frac_hbm"], 3), _r(r["evaluations_to_amortise"], 3)]
    if rp:
        line["repack@1e8"] = dict(rp, cols=["ms", "frac_hbm_after", "evals_to_amortise"])
    c5 = full.get("config5")
    if c5:
        if "error" in c5:
            line["config5"] = {"error": c5["error"][:80]}
        else:
            r = c5.get("roofline_rank0", {})
            line["config5"] = {"workload": "gv_sigma5", "value": _r(c5["value"], 5), "unit": "samples/s", "n_gpus": c5["n_gpus"], "total_samples": c5["total_samples"],
                               "steps": c5["steps"], "bound": r.get("bound"), "frac": _r(r.get("frac"), 3), "frac_hbm": _r(r.get("frac_hbm"), 3),
                               "frac_valu": _r(r.get("frac_valu"), 3), "sclk_ghz": _r(r.get("clock_ghz"), 3), "frac_power": _r(r.get("frac_power"), 3),
                               "power_w": (int(round(r["power_w"])) if r.get("power_w") else None)}
    ac = full.get("accumulate")
    if ac:
        line["accumulate"] = ({"error": ac["error"][:80]} if "error" in ac else
                              {"value": _r(ac["value"], 5), "unit": "samples/s", "kernel": ac["roofline"]["kernel"], "frac_hbm": _r(ac["roofline"]["frac_hbm"], 4),
                               "frac_valu": _r(ac["roofline"]["frac_valu"], 3)})
    mc = full.get("mc_step")
    if mc:
        line["mc_step"] = {"value": _r(mc.get("value"), 5), "unit": "samples/s", "leaf_parity": "unpinned (no Lehmann.jl)",
                           "max_dev_over_Sk": _r(mc.get("max_dev_over_Sk"), 3), "max_dev_over_Ak": _r(mc.get("max_dev_over_Ak"), 3)} if "error" not in mc else {"error": mc["error"][:80]}
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    # never let the line outgrow the driver's capture: drop the optional parts, least important first
    for k in ("repack@1e8", "mc_step", "accumulate", "secondary", "secondary_cols", "config5"):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


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
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)                      # (the same momenta and times in every run: the reported deviations are then reproducible)
        dK = torch.rand((n_loop * dim, B), dtype=torch.float64, device=dev, generator=gen) * 4 - 2        # component-major, like a Julia B x n matrix
        dT = torch.rand((n_tau, B), dtype=torch.float64, device=dev, generator=gen) * beta
        w = torch.rand(B, dtype=torch.float64, device=dev, generator=gen)
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
        # checker (outside any timed region): the first 2048 samples against the pure oracle chain -- numpy leaves from
        # (K, T), then the oracle's graph -- relative to the root's own term scale S_k (BASELINE.json's bar: 1e-12) and to
        # the absolute-value graph A_k (the first-order bound for leaves that differ in their last bit)
        dev_S = dev_A = None
        try:
            import oracle
            nchk = int(min(B, 2048))
            Kh = dK[:, :nchk].t().contiguous().cpu().numpy().reshape(nchk, n_loop, dim)
            Th = dT[:, :nchk].t().contiguous().cpu().numpy()
            h_leaf = oracle.leaf_values(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], Kh, Th, kF, beta, lam)
            want = oracle.eval_static(t, h_leaf)
            rt = torch.zeros((nchk, t.n_root), dtype=torch.float64, device=dev)
            h.mc_eval_device(dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, kF, beta, lam, rt.data_ptr(), t.n_root, 1, nchk, st)
            torch.cuda.synchronize()
            err = np.abs(rt.cpu().numpy() - want)
            dev_S = float(np.nanmax(err / np.maximum(1.0, oracle.root_scale(t, h_leaf))))
            dev_A = float(np.nanmax(err / np.maximum(1.0, oracle.abs_graph_scale(t, h_leaf))))
        except Exception as e:
            print(f"[bench] mc_step checker: {type(e).__name__}: {e}", file=sys.stderr)
        return {"value": B / ms * 1e3, "unit": "samples/s", "ms_per_call": ms, "samples_per_call": B,
                "max_dev_over_Sk": dev_S, "max_dev_over_Ak": dev_A, "checked_samples": 2048,
                "leaf_parity": "unpinned: green_derive restates Lehmann.jl's published definition (Lehmann.jl is not in the reference checkout); pinned by mpmath only",
                "what": "fdg_mc_accumulate_device: leaves from (K, T) + graph + weighted accumulation; on this handle one kernel of the "
                        "optimizing back end for programs of up to 40 000 + 30 L ops (leaves are values computed in registers), leaf kernel + evaluator above",
                "input_bytes_per_sample": 8 * (n_loop * dim + n_tau + 1), "parameters": {"kF": kF, "beta": beta, "lambda": lam},
                "parity": "graph part bit-exact given the kernel's leaves; leaves within 1e-13 relative / 1e-12 of the largest Leibniz term of the oracle's (tests/test_gpu_parity.py)"}
    except Exception as e:                      # secondary: never takes the headline line down
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline(t, case, budget_s):
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
    n1 = min(int(case.B), 4096)
    h1 = case.head(case.leaf, n1)
    cb(h1[:256], 1)
    t0 = time.perf_counter()
    cb(h1, 1)
    rate1 = n1 / max(time.perf_counter() - t0, 1e-9)
    want = max(4096, rate1 * cores * 1.0)
    n = int(min(case.B, want, 1 << 22))
    h = case.head(case.leaf, n)
    cb(h[: min(n, 64 * cores)], cores)           # thread start-up outside the clock
    t0 = time.perf_counter()
    cb(h, cores)                                 # calibration pass with all threads
    t_pass = max(time.perf_counter() - t0, 1e-6)
    reps = int(max(1, min(10000, round(budget_s / t_pass))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = cb(h, cores)
    dt = time.perf_counter() - t0
    got = case.head(case.root, n)
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
