#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/full_bench_line.json 2> gpurun_out/full_bench.err ) 2>&1 | grep real
cp bench_detail.json gpurun_out/full_bench_detail.json
python - <<PY
import json
d=json.load(open("gpurun_out/full_bench_line.json"))
print(d["config"].get("device"), d["roofline"]["frac"], d["roofline"]["frac_hbm_min_over_steps"], d["roofline"]["placement"])
for r in d["secondary"]: print(r)
print(d["config5"]); print(d["accumulate"]); print(d.get("mc_step"))
PY
wc -c gpurun_out/full_bench_line.json
