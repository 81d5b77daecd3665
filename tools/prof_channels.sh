#!/bin/bash
# Round 5: per-L2-channel (TCC instance) memory-side counters of the headline launch, several processes = several allocations.
# usage: tools/prof_channels.sh <tag> <n_processes> [bench args]      summaries -> gpurun_out/prof_<tag>/channels_*.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; NP=${2:-3}; shift 2
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=("TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_REQ TCC_BUSY" "TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL")
for p in $(seq 1 $NP); do
  g=0
  for grp in "${GROUPS_[@]}"; do
    g=$((g+1))
    D="$OUT/proc${p}_grp${g}"; mkdir -p "$D"
    timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "fdg_isa_eval" --output-format json csv -d "$D" -o p -- \
      python $R/bench.py --steps 4 --warmup 12 --no-cpu-baseline --no-secondary --no-mc-step "$@" > "$D.log" 2>&1
    python $R/tools/pmc_channels.py "$D" > "$OUT/channels_proc${p}_grp${g}.txt" 2>&1
    head -30 "$OUT/channels_proc${p}_grp${g}.txt"
    # keep one small raw sample for the schema, drop the rest (large)
    if [ "$p" = 1 ] && [ "$g" = 1 ]; then for j in $(find "$D" -name "*.json"); do head -c 300000 "$j" > "$OUT/raw_sample.json.head"; python - "$j" > "$OUT/raw_schema.txt" 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def walk(x, pre="", depth=0):
    if depth > 7: return
    if isinstance(x, dict):
        for k, v in x.items():
            print(f"{pre}{k}: {type(v).__name__}" + (f" len={len(v)}" if isinstance(v, (list, dict, str)) else f" = {v}"))
            walk(v, pre + "  ", depth + 1)
    elif isinstance(x, list) and x:
        walk(x[0], pre + "[0] ", depth + 1)
walk(d)
PY
    done; fi
    find "$D" -name "*.json" -size +1M -delete
  done
done
