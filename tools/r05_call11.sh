#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_batch_alloc.py -q -m gpu 2>&1 | tail -3
for lay in tile_major leaf_major; do
  timeout 300 python bench.py --workload gv_sigma5 --layout $lay --placement plain --steps 40 --warmup 60 --no-cpu-baseline --no-secondary --no-mc-step > gpurun_out/r05_gv5_$lay.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_gv5_$lay.json")); r=d["roofline"]
print("gv_sigma5 $lay", r["frac"], r.get("clock_ghz"), r.get("kernel"))
PY
done
timeout 900 python bench.py > gpurun_out/r05_e_bench_line.json 2> gpurun_out/r05_e_bench.err
cp bench_detail.json gpurun_out/r05_e_bench_detail.json
python - <<PY
import json
d=json.load(open("gpurun_out/r05_e_bench_line.json"))
print(d["roofline"]["frac"], d["roofline"]["frac_hbm_min_over_steps"], d["roofline"]["placement"])
for r in d["secondary"]: print(r)
print(d["config5"]); print(d["accumulate"])
PY
wc -c gpurun_out/r05_e_bench_line.json
