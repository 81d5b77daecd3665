#!/bin/bash
# round 5 (dev build, timing only: results are garbage): what would hiding the memory latency of a tile's opening loads buy a one-wave kernel?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export FDG_LIBRARY=$R/feynmandiagram.jl_amd/lib/libfdg_dev.so
O=gpurun_out/nohead.txt; : > $O
run() { timeout 600 python tools/gpu_option_sweep.py $1 $2 - FDG_ISA_DEBUG=nohead - FDG_ISA_DEBUG=nohead 2>&1 | grep -v "Warning\|amdgpu.ids" >> $O; }
run parquet_sigma4_insdyn 4000000
run parquet_sigma5 4000000
run parquet_sigma4_dyn 8000000
run gv_sigma4_taylor2 8000000
run parquet_ver4_4 1000000
run gv_sigma5 4000000
cat $O
