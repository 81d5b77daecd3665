#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() { # tag, env flags, extra args
  FDG_BENCH_PAIR_FLAGS=$2 timeout 300 python bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-secondary --no-mc-step $3 > gpurun_out/r05_b_$1.json 2> gpurun_out/r05_b_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_b_$1.json")); r=d["roofline"]
print("$1", r["frac"], r.get("frac_hbm_min_over_steps"), r.get("placement"), d.get("accumulate",{}).get("frac_hbm"))
PY
}
for i in 1 2 3 4 5 6; do run after_$i 0 ""; done
run plain_1 0 "--placement plain"; run plain_2 0 "--placement plain"
