#!/bin/bash
# usage: tools/prof_mem.sh <outdir-under-gpurun_out> <command...>
# (every pass under `timeout`: a counter group rocprofv3 cannot schedule aborts and then hangs)
# Memory-path PMC passes (one counter group per run; --kernel-trace only, as the GPU pool requires),
# then per-kernel averages.  Used to compare the evaluator with the column-stream micro-benchmark.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum" \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  ( cd "$R" && timeout 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- "$@" > "$OUT/pass$i.log" 2>&1 )
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
kern = sorted({k[0] for k in acc})
for kn in kern:
    print("kernel", kn)
    for (k, c), (v, n) in sorted(acc.items()):
        if k == kn: print("   %-44s avg/dispatch %.4g   (n=%d)" % (c, v / n, n))
PY
