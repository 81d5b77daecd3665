#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python tools/gpu_pair_settle_probe.py 10 0 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_settle.log
cat gpurun_out/r05_settle.log
