#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-secondary --no-mc-step > gpurun_out/r05_paired_$i.json 2> gpurun_out/r05_paired_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_paired_$i.json")); r=d["roofline"]
print("paired $i", d["value"], r["frac"], r.get("frac_hbm_min_over_steps"), r.get("placement"), d.get("accumulate",{}).get("frac_hbm"))
PY
done
for i in 1 2; do
  timeout 300 python bench.py --placement plain --steps 40 --warmup 60 --no-cpu-baseline --no-secondary --no-mc-step > gpurun_out/r05_plain_$i.json 2> gpurun_out/r05_plain_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_plain_$i.json")); r=d["roofline"]
print("plain $i", d["value"], r["frac"], r.get("frac_hbm_min_over_steps"), r.get("placement"), d.get("accumulate",{}).get("frac_hbm"))
PY
done
grep -h "fdg_batch_alloc_pair\|Error\|error" gpurun_out/r05_paired_*.err | head
