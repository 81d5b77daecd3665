#!/bin/bash
# round 5 (dev build): how much does the mere position of a kernel's code matter?  n s_nop at the head of the kernel (FDG_ISA_SHIFT) against
# the pads of FDG_ISA_ALIGN=1 (every 8-byte instruction) / 2 (v_ instructions only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export FDG_LIBRARY=$R/feynmandiagram.jl_amd/lib/libfdg_dev.so
O=gpurun_out/align_shift.txt; : > $O
run() { timeout 900 python tools/gpu_option_sweep.py $1 $2 FDG_ISA_ALIGN=0 FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=1 FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=2 FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=3 FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=4 \
   FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=8 FDG_ISA_ALIGN=0,FDG_ISA_SHIFT=15 FDG_ISA_ALIGN=1 FDG_ISA_ALIGN=1,FDG_ISA_SHIFT=1 FDG_ISA_ALIGN=1,FDG_ISA_SHIFT=8 FDG_ISA_ALIGN=2 FDG_ISA_ALIGN=0 2>&1 | grep -v "Warning\|amdgpu.ids" >> $O; }
run parquet_sigma4_dyn 8000000
run gv_sigma4 16000000
run gv_sigma4_taylor2 8000000
run gv_ver4_4 500000
run parquet_sigma4_insdyn 4000000
cat $O
