#!/bin/bash
# usage: tools/prof_mc.sh <tag> [workload] [n_sample]  -- kernel-trace stats + PMC passes of the Monte-Carlo step (routes split and isa)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; W=${2:-gv_sigma4_taylor2}; N=${3:-4000000}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $R/tools/gpu_mc_isa_check.py $W $N > "$OUT/run.txt" 2> "$OUT/trace.log"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- python $R/tools/gpu_mc_isa_check.py $W $N > "$OUT/pass$i.log" 2>&1
done
python $R/tools/rocpd_stats.py $(find "$OUT/trace" -name "*.db") > "$OUT/kernel_stats.txt" 2>&1
python $R/tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/run.txt"; cat "$OUT/kernel_stats.txt"; head -40 "$OUT/pmc_summary.txt"
