#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args]   -- kernel-trace stats + PMC passes of one bench.py run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $R/bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/trace.log"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQC_ICACHE_MISSES" \
  "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- python $R/bench.py --steps 3 --warmup 30 --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1
done
python $R/tools/rocpd_stats.py --last=${STEPS:-100} $(find "$OUT/trace" -name "*.db") > "$OUT/kernel_stats.txt" 2>&1
python $R/tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/kernel_stats.txt"; head -40 "$OUT/pmc_summary.txt"; cat "$OUT/bench.json" | cut -c1-300
