#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 600 python tools/gpu_pair_alloc_probe.py parquet_sigma4 100000000 1 0 cal > gpurun_out/r05_pair_alloc_dbg_$i.log 2>&1
cat gpurun_out/r05_pair_alloc_dbg_$i.log
done
