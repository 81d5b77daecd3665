#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun): tools/prof_r06_set.sh <tag> ["w1 layout" "w2 layout" ...]
#  1. rocprofv3 --kernel-trace --stats of `python bench.py` (headline alone: batch from fdg_batch_alloc_pair, so the trace also holds the
#     allocator's short probe launches of the same kernel -- the summary reports the LAST 100 launches, bench.py's timed steps, separately)
#  2. per (workload, layout), separate --pmc passes FETCH_SIZE | WRITE_SIZE of `bench.py --workload W --layout LAY --placement plain`
#     (plain allocation: every dispatch of the kernel is then a full-batch launch; the traffic does not depend on where the pages lie)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_headline" -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --no-mc-step > "$OUT/bench_headline.json" 2> "$OUT/trace_headline.log"
python $R/tools/rocpd_stats.py --last=100 $(find "$OUT/trace_headline" -name "*.db") > "$OUT/kernel_stats_headline.txt" 2>&1
if [ $# -eq 0 ]; then set -- "parquet_sigma4 tile_major"; fi
for spec in "$@"; do
  set -- $spec
  W=$1; LAY=$2
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    D="$OUT/pmc_${W}_${LAY}/pass$i"
    mkdir -p "$D"
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$D" -o p -- python $R/bench.py --workload $W --layout $LAY --placement plain --steps 3 --warmup 30 --no-cpu-baseline --no-secondary --no-mc-step > "$D.log" 2>&1
  done
  python $R/tools/pmc_summary.py "$OUT/pmc_${W}_${LAY}" > "$OUT/pmc_${W}_${LAY}.txt" 2>&1
done
python $R/tools/make_traffic_json.py "$OUT" "$OUT/traffic.json" r06 > /dev/null 2>&1
find "$OUT" -name "*counter_collection.csv" -size +2M -delete; find "$OUT" -name "*.db" -size +20M -delete
cat "$OUT/kernel_stats_headline.txt" | cut -c1-200; cut -c1-400 "$OUT/bench_headline.json"
