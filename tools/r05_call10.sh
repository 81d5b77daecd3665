#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests.log 2>&1
tail -5 gpurun_out/r05_gputests.log
timeout 900 python bench.py > gpurun_out/r05_c_bench_line.json 2> gpurun_out/r05_c_bench.err
cp bench_detail.json gpurun_out/r05_c_bench_detail.json
cut -c1-1500 gpurun_out/r05_c_bench_line.json
