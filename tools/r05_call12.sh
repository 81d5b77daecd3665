#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
show() { python - <<PY
import json
d=json.load(open("$1"))
print("$1", d["roofline"]["frac"], d["roofline"]["frac_hbm_min_over_steps"])
for r in d.get("secondary", []): print("  ", r)
PY
}
timeout 600 python bench.py --steps 20 --warmup 60 --no-cpu-baseline --no-mc-step --secondary gv_sigma5:tile_major,gv_sigma5:leaf_major,gv_sigma5:tile_major,gv_sigma5:leaf_major,gv_sigma5:tile_major > gpurun_out/r05_f1.json 2>/dev/null; show gpurun_out/r05_f1.json
timeout 900 python bench.py --steps 20 --warmup 60 --no-cpu-baseline --no-mc-step --pair-all --secondary gv_sigma5:tile_major,gv_sigma4_taylor2:tile_major,gv_sigma4_taylor2:sample_major,parquet_sigma4_taylor2:tile_major,parquet_sigma5:tile_major,parquet_sigma4_insdyn:tile_major > gpurun_out/r05_f2.json 2>/dev/null; show gpurun_out/r05_f2.json
