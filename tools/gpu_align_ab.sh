#!/bin/bash
# round 5: kernels printed with / without the alignment pads (FDG_ISA_ALIGN), A B A B in one process per graph
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/align_ab.txt; : > $O
run() { timeout 600 python tools/gpu_option_sweep.py $1 $2 FDG_ISA_ALIGN=0 - FDG_ISA_ALIGN=0 - 2>&1 | grep -v Warning >> $O; }
run parquet_sigma4_insdyn 4000000
run parquet_sigma5 4000000
run gv_sigma5 4000000
run parquet_sigma4_taylor2 8000000
run gv_sigma4_taylor2 8000000
run parquet_ver4_4 1000000
run gv_ver4_4 500000
run parquet_sigma4_dyn 8000000
run parquet_sigma4 16000000
run gv_sigma4 16000000
cat $O
