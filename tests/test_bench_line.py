"""bench.py's stdout contract: ONE JSON line short enough to survive the driver's capture (it keeps the
last ~8 KB of stdout; round 2's 21 KB line lost its head and with it value / roofline / cpu_baseline), details in
bench_detail.json; config 5 sized by BASELINE.json's 10^9 samples; and the N > 1 path of bench.py itself run as a dry
run -- one process per rank on the CPU over gloo, the evaluator replaced by a stub -- so that sharding, the single
collective and the line are exercised without a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned(n_secondary=15):
    long = "x" * 400
    roof = {"bound": "hbm", "achieved": 5790.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.7237654321, "traffic": 70400000000.0,
            "traffic_over_algorithmic": 1.0000001, "traffic_source": long, "traffic_source_detail": long, "kernel": "fdg_isa_eval_nt",
            "avg_kernel_ms": 12.1547, "frac_hbm": 0.7237654321, "frac_valu": 0.3512345, "frac_hbm_min_over_steps": 0.70123, "frac_hbm_max_over_steps": 0.74,
            "measured_copy_gbs": 5961.2, "frac_of_measured_copy": 0.97, "ops_exec_per_eval": 1173, "algorithmic_bytes_per_launch": 70400000000, "valu_tops": 9.6}
    sec = [{"workload": "parquet_sigma4_insdyn " + long, "layout": "sample_major" if i % 3 == 0 else "leaf_major", "value": 1.234567e9 * (i + 1), "unit": "evals/s",
            "roofline": dict(roof, bound="valu_fp64" if i % 2 else "hbm"), "gpu_matches_cpu_bitwise": True, "max_abs_dev": 0.0, "kernel_info": {"a": 1}}
           for i in range(n_secondary)]
    sec.append({"workload": "broken", "layout": "leaf_major", "error": "RuntimeError: " + long})
    return {"metric": "graph-evaluations/sec", "value": 8.2273456789e9, "unit": "evals/s", "n_gpus": 1, "steps": 100, "warmup": 60, "ms_per_step": 12.154789,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "parquet_sigma4" + bench.WORKLOAD_NOTES["parquet_sigma4"], "graph": "g", "n_leaf": 84, "n_node": 1325, "n_edge": 2855, "n_root": 4,
                       "flops_per_eval": 1, "bytes_per_eval": 704, "samples_per_step_per_gpu": 100000000, "layout": "leaf_major", "settle_steps": 0,
                       "kernel": long, "parallelism": "samples sharded x1, one all-reduce of 4 doubles", "parity": bench.PARITY_NOTE},
            "roofline": roof, "valu_fp64": {"note": long}, "kernel_info": {"max_live": 1},
            "cpu_baseline": {"value": 7.7e7, "unit": "evals/s", "cores": 256, "kind": "port", "sample": long, "gpu_matches_cpu_bitwise": True, "max_abs_dev": 0.0},
            "config5": {"workload": "gv_sigma5" + long, "value": 1.5e9, "unit": "samples/s (whole job)", "n_gpus": 1, "steps": 500, "total_samples": 1e9,
                        "roofline_rank0": dict(roof, bound="valu_fp64"), "observable": [1.0, 2.0], "what": long},
            "accumulate": {"value": 9.38e9, "unit": "samples/s", "roofline": dict(roof, kernel="fdg_isa_eval_acc_nt", frac_hbm=0.797), "what": long},
            "secondary": sec, "secondary_note": long,
            "mc_step": {"value": 7.2e9, "unit": "samples/s", "what": long, "max_dev_over_Sk": 4.4e-13, "max_dev_over_Ak": 1.6e-15, "parity": long}}


def test_line_is_short_and_keeps_the_contract_keys():
    line = bench.compact_line(_canned())
    assert len(line) < bench.LINE_LIMIT <= 4000 and "\n" not in line
    got = json.loads(line[-8000:])                      # what the driver does with the tail of stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in got, k
    assert set(got["config"]) >= {"workload", "layout", "samples_per_step_per_gpu"} and "model" not in got["config"] and len(got["config"]["workload"]) <= 120
    r = got["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s" and r["traffic"] == 70400000000.0
    assert r["kernel"] == "fdg_isa_eval_nt" and len(r["traffic_source"]) <= 30 and 0.70 <= r["frac_hbm_min_over_steps"] <= r["frac"]
    c = got["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 256 and c["value"] > 0 and len(c["sample"]) <= 110
    assert len(got["secondary"]) == 16 and all(len(row) == len(got["secondary_cols"]) for row in got["secondary"])
    assert got["secondary"][0][0] == "parquet_sigma4_insdyn" and got["secondary"][0][1] == "rm" and got["secondary"][1][3] == "valu"
    assert got["secondary"][-1][3] == "error"
    assert got["config5"]["total_samples"] == 1e9 and got["config5"]["bound"] == "valu_fp64"
    assert got["accumulate"]["kernel"] == "fdg_isa_eval_acc_nt" and got["accumulate"]["frac_hbm"] == 0.797 and got["accumulate"]["value"] == 9.38e9
    assert got["mc_step"]["leaf_parity"].startswith("unpinned") and got["mc_step"]["max_dev_over_Sk"] == 4.4e-13
    assert abs(got["value"] - 8.2273456789e9) / 8.2e9 < 1e-5


def test_line_sheds_optional_parts_rather_than_outgrow_the_limit():
    line = bench.compact_line(_canned(n_secondary=80))
    assert len(line) < bench.LINE_LIMIT
    got = json.loads(line)
    assert "roofline" in got and "cpu_baseline" in got and "secondary" not in got


def test_binding_roof_is_the_larger_minimum_time():
    st = {"bytes_alg": 8 * (84 + 4), "bytes_alg_accumulate": 8 * 84}
    r = bench.roofline_of(st, 10**8, 12.15e-3, "k", ops_exec=1173)
    assert r["bound"] == "hbm" and abs(r["frac"] - 0.7243) < 1e-3 and abs(r["frac_valu"] - 1173e8 / 12.15e-3 / 39.3e12) < 1e-6
    st5 = {"bytes_alg": 8 * 359, "bytes_alg_accumulate": 8 * 357}
    r5 = bench.roofline_of(st5, 2 * 10**6, 1.3e-3, "k", ops_exec=20000)        # 7 fold steps per byte: above the ridge (4.9 op/B)
    assert r5["bound"] == "valu_fp64" and r5["unit"] == "TFLOP/s" and r5["peak"] == 39.3 and r5["frac"] == r5["frac_valu"] > r5["frac_hbm"]
    assert bench.roofline_of(st, 10, 1.0, "k")["bound"] == "hbm"               # no op count: the HBM roof


def test_config5_is_sized_by_baseline_not_by_steps():
    assert bench.CONFIG5_TOTAL_SAMPLES == 10**9
    assert [bench.config5_steps(w) for w in (1, 2, 4, 8)] == [500, 250, 125, 63]
    assert all(bench.config5_steps(w) * w * bench.DEFAULT_B["gv_sigma5"] >= 10**9 for w in (1, 2, 3, 4, 8))


@pytest.mark.parametrize("world", [1, 2])
def test_dry_run_of_bench_itself(world, tmp_path):
    """`bench.py --gpus N --dry-run` under torch.distributed.run exactly as the driver launches it (gloo instead of RCCL, a
    stub instead of the evaluator whose accumulate adds the shard's sample count): rank 0 prints one parseable line, the
    one collective of config 5 sums to the job's 10^9+ samples, shards are the contiguous ranges of sharding.shard_range."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run"]
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29620 + world)] + cmd[1:]
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 4000
    got = json.loads(lines[0])
    assert got["n_gpus"] == world and got["steps"] == 3 and got["scaling"] == "weak" and got["data"].startswith("dry-run")
    c5 = got["config5"]
    assert c5["n_gpus"] == world and c5["steps"] == bench.config5_steps(world)
    assert c5["total_samples"] == c5["steps"] * world * bench.DEFAULT_B["gv_sigma5"] >= 10**9
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    # every rank added its shard's sample count once per step; ONE all-reduce summed the ranks
    assert detail["config5"]["observable"][0] == c5["total_samples"]
    assert detail["config"]["shard_offset_rank0"] == 0 and detail["config5"]["shard_offset_rank0"] == 0


def test_dry_run_with_eight_ranks_is_config_5_as_baseline_words_it(tmp_path):
    """BASELINE.json configs[4]: 10^9 samples sharded across 8 GPUs with one final reduce.  The driver's own command line for
    N = 8, on the CPU: 63 steps of 8 x 2*10^6 samples (10^9 rounded UP to whole steps: 1.008*10^9, said so in the line), eight
    contiguous shards of a step in rank order, and the single collective sums every rank's samples."""
    world = 8
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29637", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--dry-run"]
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    got = json.loads(lines[0])
    assert got["n_gpus"] == 8 and got["scaling"] == "weak"
    c5 = got["config5"]
    per = bench.DEFAULT_B["gv_sigma5"]
    assert c5["steps"] == 63 and c5["n_gpus"] == 8 and c5["total_samples"] == 63 * 8 * per == 1_008_000_000 >= 10**9
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))["config5"]
    assert detail["shards_of_a_step"] == [[r * per, per] for r in range(8)]
    assert detail["observable"][0] == 1_008_000_000.0            # the one all-reduce: every rank added its shard once per step
    assert detail["baseline_total_samples"] == 10**9


@pytest.mark.gpu
def test_bench_line_on_the_device(tmp_path):
    """The real thing, shortened: `python bench.py` on cuda:0 with a smaller batch and two secondary rows.  One parseable
    line; the kernel named in it is the one the library launched (hand-written ISA, from fdg_graph_kernel_info); roofline
    and cpu_baseline present and consistent; every checked sample bit-equal to the CPU port."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "3", "--samples", "4000000", "--cpu-seconds", "2",
           "--secondary", "parquet_sigma4:sample_major,gv_sigma5:leaf_major,parquet_sigma4:leaf_major"]
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT, p.stdout[-2000:]
    got = json.loads(lines[0][-8000:])
    assert got["metric"] == "graph-evaluations/sec" and got["n_gpus"] == 1 and got["steps"] == 5 and got["dtype"] == "f64" and got["vs_baseline"] is None
    assert got["config"]["workload"].startswith("parquet_sigma4") and got["config"]["samples_per_step_per_gpu"] == 4000000
    assert got["config"]["layout"] == "tile_major" and got["roofline"]["placement"].startswith("fdg_batch_alloc_pair")
    r = got["roofline"]
    assert r["kernel"] == "fdg_isa_eval_nt" and r["bound"] == "hbm" and r["ops_exec_per_eval"] > 0
    assert abs(r["achieved"] - got["value"] * 704 / 1e9) / r["achieved"] < 0.1          # 8 (L + R) bytes per evaluation, HIP events vs wall clock
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3 and r["frac_hbm_min_over_steps"] <= r["frac"] + 1e-9
    assert got["value"] > 1e9 and r["frac"] > 0.3                                        # (a smaller batch than the default: not the headline's figure)
    assert r["measured_read_gbs"] > 3000 and 0.3 < r["frac_of_measured_read"] < 1.2          # fdg_read_device: the memory system's ceiling for a read stream on this box
    assert r.get("power_w") is None or (200 < r["power_w"] <= 1.05 * (r.get("power_cap_w") or 1400))        # rocm-smi next to the headline launch, when there is one
    assert 0.3 < r["frac_power"] < 1.3 and "power_w" in got["secondary_cols"]            # the power roof for the headline; the rows carry what rocm-smi read next to them
    c = got["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_matches_cpu_bitwise"] is True
    rows = {(x[0], x[1]): x for x in got["secondary"]}
    i_bit, i_clk, i_pw = got["secondary_cols"].index("bitwise"), got["secondary_cols"].index("sclk_ghz"), got["secondary_cols"].index("power_w")
    # (a star: the row's batch came from fdg_batch_alloc_pair; the parquet_sigma4 rows do, gv_sigma5's does not)
    assert set(rows) == {("parquet_sigma4", "rm*"), ("gv_sigma5", "lm"), ("parquet_sigma4", "lm*")} and all(x[i_bit] == 1 for x in rows.values())
    # rocm-smi next to the row's launches (round 6; the sleeping-wave probe of round 5 stays in the detail file): a plausible shader clock and socket power
    assert all(x[i_clk] is None or 1.2 < x[i_clk] < 2.6 for x in rows.values()) and (r.get("clock_ghz") is None or 1.2 < r["clock_ghz"] < 2.6)
    assert all(x[i_pw] is None or 300 < x[i_pw] < 1500 for x in rows.values())
    assert any(x[i_pw] is not None for x in rows.values()) or r.get("power_w") is None      # (no rocm-smi on the box: no power anywhere)
    assert got["config5"]["total_samples"] >= 10**9 and got["config5"]["n_gpus"] == 1
    assert got["accumulate"]["kernel"].startswith("fdg_isa_eval_acc") and got["accumulate"]["value"] > got["value"] * 0.8
    assert got["mc_step"]["value"] > 0 and got["mc_step"]["max_dev_over_Sk"] < 1e-11 and got["mc_step"]["max_dev_over_Ak"] < 1e-14
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert detail["secondary"][0]["roofline"]["kernel"] == "fdg_isa_eval_rl"        # contiguous rows: the linear row-major variant
    assert detail["config5"]["roofline_rank0"]["kernel"].startswith("fdg_isa_eval_acc")
