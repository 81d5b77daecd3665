cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 600 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_n.txt; }
: > gpurun_out/r06_log_sweep_n.txt
export SWEEP_LAYOUT=rm
echo "row-major: chunks fetched in pairs (four buffers = two pairs) so that the line two chunks of a row share is asked for twice within nanoseconds" | tee -a gpurun_out/r06_log_sweep_n.txt
run gv_sigma4_taylor2 4000000 - FDG_ISA_RM_BUFS=4 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1,FDG_ISA_DEBUG=novalu -
run parquet_sigma5 2000000 - FDG_ISA_RM_BUFS=4 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1,FDG_ISA_DEBUG=novalu -
run parquet_sigma4_insdyn 2000000 - FDG_ISA_RM_BUFS=4 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 -
run gv_sigma5 2000000 - FDG_ISA_RM_BUFS=4 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 -
run parquet_sigma4_dyn 4000000 - FDG_ISA_RM_BUFS=4 FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 -
run parquet_sigma4_taylor2 4000000 - FDG_ISA_RM_WAVES=1,FDG_ISA_RM_BUFS=4 FDG_ISA_RM_WAVES=1,FDG_ISA_RM_BUFS=4,FDG_RM_PAIR=1 -
