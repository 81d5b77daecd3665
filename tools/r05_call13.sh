#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_tile_major.py tests/test_random_graphs.py tests/test_full_size.py -q -m gpu 2>&1 | tail -3
show() { python - <<PY
import json
d=json.load(open("$1"))
print("$1", d["roofline"]["frac"], d["roofline"]["kernel"])
for r in d.get("secondary", []): print("  ", r)
PY
}
SEC=parquet_sigma4_insdyn:tile_major,parquet_sigma5:tile_major,parquet_ver4_4:tile_major,gv_ver4_4:tile_major,gv_sigma5:tile_major
timeout 600 python bench.py --workload parquet_sigma4_insdyn --placement plain --steps 40 --warmup 60 --no-cpu-baseline --no-mc-step --secondary $SEC > gpurun_out/r05_g1.json 2>gpurun_out/r05_g1.err; show gpurun_out/r05_g1.json
grep -o '"kernel": "[a-z_0-9]*"' gpurun_out/r05_g1.err | sort | uniq -c
