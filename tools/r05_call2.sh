#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python tools/gpu_pair_probe.py parquet_sigma4 100000000 malloc 3 2 > gpurun_out/r05_pair_probe.log 2>&1
cat gpurun_out/r05_pair_probe.log
