#!/bin/bash
# round 5, GPU call 1: (a) is the headline's rate local to the pages?  (b) per-L2-channel counters of a few allocations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python tools/gpu_chunk_probe.py parquet_sigma4 100000000 malloc,whole,1024,2 2.0 2 > gpurun_out/r05_chunk_probe.log 2>&1
tail -30 gpurun_out/r05_chunk_probe.log
timeout 1500 tools/prof_channels.sh r05_channels 3 > gpurun_out/r05_channels.log 2>&1
tail -60 gpurun_out/r05_channels.log
