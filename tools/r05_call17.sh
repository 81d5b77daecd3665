#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -14
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-mc-step --no-cpu-baseline > gpurun_out/r05_j_bench_line_$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r05_j_bench_line_$i.json")); r=d["roofline"]
print(d["config"].get("device"), d["value"], r["frac"], r["frac_hbm_min_over_steps"], r["placement"], d["accumulate"]["frac_hbm"])
PY
done
