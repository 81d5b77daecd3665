#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/gpu_pair_matrix.py parquet_sigma4 100000000 malloc 2.0 1 > gpurun_out/r05_pair_matrix.log 2>&1
timeout 600 python tools/gpu_pair_matrix.py parquet_sigma4 100000000 whole 2.0 1 >> gpurun_out/r05_pair_matrix.log 2>&1
timeout 600 python tools/gpu_pair_matrix.py parquet_sigma4 100000000 1024 1.0 1 >> gpurun_out/r05_pair_matrix.log 2>&1
cat gpurun_out/r05_pair_matrix.log
