#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python tools/gpu_pair_matrix2.py parquet_sigma4 100000000 1008 48 64 malloc > gpurun_out/r05_pair_matrix2.log 2>&1
timeout 900 python tools/gpu_pair_matrix2.py parquet_sigma4 100000000 336 16 64 malloc >> gpurun_out/r05_pair_matrix2.log 2>&1
cat gpurun_out/r05_pair_matrix2.log
