#!/bin/bash
# usage: tools/prof_alloc.sh <tag>  -- per-dispatch TLB / stall counters of the headline launch over several placements of its matrix
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-alloc}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python $R/tools/gpu_alloc_probe.py parquet_sigma4 100000000 8 > "$OUT/plain.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
# (PROF_ALLOC_GROUPS="grp1|grp2|...": other counter groups, e.g. the memory-side request counters)
IFS='|' read -r -a GROUPS_ <<< "${PROF_ALLOC_GROUPS:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum|TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_UTCL2_BUSY}"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  D="$OUT/pass$i"; mkdir -p "$D"
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$D" -o p -- python $R/tools/gpu_alloc_probe.py parquet_sigma4 100000000 6 > "$D.log" 2>&1
  python - "$D" <<'PY' > "$OUT/pass$i.txt" 2>&1
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fdg_isa_eval"):
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fdg_isa_eval"):
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in cnt.values() for k in v})
print("dispatch  ms  " + "  ".join(names))
for k in sorted(cnt, key=lambda x: int(x)):
    print(k, "%.3f" % dur.get(k, -1), "  ".join("%.4g" % cnt[k].get(n, -1) for n in names))
PY
done
cat "$OUT/plain.txt"; grep "round" "$OUT"/pass*.log; for f in "$OUT"/pass*.txt; do echo "== $f"; head -60 "$f"; done
