#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python tools/gpu_pair_alloc_probe.py parquet_sigma4 100000000 4 0 cal,malloc > gpurun_out/r05_pair_alloc.log 2>&1
cat gpurun_out/r05_pair_alloc.log
