#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gputests.log 2>&1
tail -5 gpurun_out/gputests.log
timeout 900 python bench.py --no-secondary --no-mc-step --no-cpu-baseline > gpurun_out/bench_line_headline_only.json 2> gpurun_out/bench_headline.err

cut -c1-900 gpurun_out/bench_line_headline_only.json
