#!/bin/bash
# usage: tools/prof_counters.sh <tag> <workload> "<counters>" [more counter groups]   (env FDG_* passes through; LAYOUT=sample_major for row-major input)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; W=$2; shift 2
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  D="$OUT/pass$i"; mkdir -p "$D"
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$D" -o p -- python $R/bench.py --workload $W --layout ${LAYOUT:-leaf_major} --steps 3 --warmup 10 --no-cpu-baseline --no-secondary --no-mc-step > "$D.log" 2>&1
done
python $R/tools/pmc_summary.py "$OUT" | grep -A40 "fdg_isa_eval" | head -44
