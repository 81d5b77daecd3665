#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes for a list of "workload layout" pairs: tools/prof_pmc_list.sh <tag> "w1 leaf_major" "w2 sample_major" ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  W=$1; LAY=$2
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D="$OUT/pmc_${W}_${LAY}/pass$i"
    mkdir -p "$D"
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$D" -o p -- python $R/bench.py --workload $W --layout $LAY --steps 3 --warmup 30 --no-cpu-baseline --no-secondary --no-mc-step > "$D.log" 2>&1
  done
  python $R/tools/pmc_summary.py "$OUT/pmc_${W}_${LAY}" > "$OUT/pmc_${W}_${LAY}.txt" 2>&1
  head -4 "$OUT/pmc_${W}_${LAY}.txt"
done
