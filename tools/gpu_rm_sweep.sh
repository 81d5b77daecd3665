#!/bin/bash
# round 5 (dev build): the chunked row-major variant's leaf-read lookahead (FDG_ISA_RM_LA, default 48) on the graphs with rows longer than 98 leaves
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
export FDG_LIBRARY=$PWD/feynmandiagram.jl_amd/lib/libfdg_dev.so FDG_CACHE_DIR=/tmp/rmcache
run() { w=$1; shift; env "$@" python bench.py --workload $w --layout sample_major --placement plain --steps 20 --warmup 30 --no-cpu-baseline --no-secondary --no-mc-step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $*', r['kernel'], round(r['frac_hbm'],4), round(d['value']/1e9,3))"; }
for w in ${WL:-gv_sigma4_taylor2 parquet_sigma5 parquet_sigma4_insdyn gv_sigma5 parquet_sigma4_dyn parquet_sigma4_taylor2}; do
  for la in 48 72 96 128 48; do run $w FDG_ISA_RM_LA=$la; done
done 2>&1 | tee gpurun_out/rm_sweep.txt
