#!/bin/bash
# usage: tools/prof_one.sh <tag> <workload> [layout]  -- FETCH_SIZE / WRITE_SIZE / SQ passes of one workload's evaluator kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; W=$2; LAY=${3:-leaf_major}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  D="$OUT/pass$i"; mkdir -p "$D"
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$D" -o p -- python $R/bench.py --workload $W --layout $LAY --steps 3 --warmup 10 --no-cpu-baseline --no-secondary --no-mc-step > "$D.log" 2>&1
done
python $R/tools/pmc_summary.py "$OUT" | head -30
