#!/bin/bash
# round 5 (dev build, timing only: results are garbage): a one-wave kernel without its AGPR moves / LDS slots / leaf loads / arithmetic
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export FDG_LIBRARY=$R/feynmandiagram.jl_amd/lib/libfdg_dev.so
O=gpurun_out/onewave_split.txt; : > $O
run() { timeout 600 python tools/gpu_option_sweep.py $1 $2 - FDG_ISA_DEBUG=noacc FDG_ISA_DEBUG=nolds FDG_ISA_DEBUG=noacc+nolds FDG_ISA_DEBUG=noleaf FDG_ISA_DEBUG=noleaf+noacc+nolds FDG_ISA_DEBUG=novalu FDG_ISA_DEBUG=novalu+noacc+nolds FDG_ISA_DEBUG=novmwait FDG_ISA_DEBUG=novmwait+noacc+nolds - 2>&1 | grep -v "Warning\|amdgpu.ids" >> $O; }
run parquet_sigma4_insdyn 4000000
run parquet_sigma5 4000000
run parquet_ver4_4 1000000
cat $O
