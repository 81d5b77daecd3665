#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun): tools/prof_r04_set.sh <tag>
#  1. rocprofv3 --kernel-trace --stats of `python bench.py` (a) headline alone, (b) the whole default run
#  2. per (workload, layout) of the bench line, separate --pmc passes FETCH_SIZE | WRITE_SIZE of `bench.py --workload W --layout LAY`
# Summaries land in gpurun_out/prof_<tag>/ ; tools/make_traffic_json.py turns the PMC passes into profiles/r04_traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_headline" -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --no-mc-step > "$OUT/bench_headline.json" 2> "$OUT/trace_headline.log"
python $R/tools/rocpd_stats.py --last=100 $(find "$OUT/trace_headline" -name "*.db") > "$OUT/kernel_stats_headline.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $R/bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/trace.log"
python $R/tools/rocpd_stats.py $(find "$OUT/trace" -name "*.db") > "$OUT/kernel_stats.txt" 2>&1
[ "${PROF_TRACE_ONLY:-0}" = 1 ] && { cat "$OUT/kernel_stats_headline.txt" "$OUT/kernel_stats.txt"; exit 0; }
SPECS=("parquet_sigma4 tile_major" "parquet_sigma4 leaf_major" "parquet_sigma4 sample_major" "parquet_sigma4_dyn tile_major" "parquet_sigma4_insdyn tile_major"
       "parquet_sigma4_taylor2 tile_major" "parquet_sigma4_taylor2 leaf_major" "parquet_sigma5 tile_major" "parquet_ver4_4 tile_major" "gv_ver4_4 tile_major" "gv_ver4_4 leaf_major"
       "sigma2 tile_major" "sigma4_standin leaf_major" "gv_sigma4 tile_major" "gv_sigma5 tile_major" "gv_sigma5 leaf_major" "gv_sigma6 leaf_major" "gv_sigma4_taylor2 tile_major" "gv_sigma4_taylor2 sample_major")
for spec in "${SPECS[@]}"; do
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
  # (the raw csv files are large: keep the summaries)
  rm -rf "$OUT/pmc_${W}_${LAY}"/pass*/p_counter_collection.csv.bak 2>/dev/null
done
cat "$OUT/kernel_stats_headline.txt"; cut -c1-300 "$OUT/bench.json"
