cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_g.txt; }
: > gpurun_out/r06_log_sweep_g.txt
export SWEEP_LAYOUT=rm
echo "row-major: two waves per SIMD (two staging buffers each) for rows longer than 128 leaves?" | tee -a gpurun_out/r06_log_sweep_g.txt
run gv_sigma4_taylor2 4000000 - FDG_ISA_RM_WAVES=2 FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=2 -
run parquet_sigma5 2000000 - FDG_ISA_RM_WAVES=2 FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=2 FDG_ISA_RM_BUFS=3 -
run parquet_sigma4_insdyn 2000000 - FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=3 -
run gv_sigma5 2000000 - FDG_ISA_RM_WAVES=2 FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=3 FDG_RM_VN=400 -
run parquet_sigma4_taylor2 4000000 - FDG_ISA_RM_WAVES=1 -
run gv_sigma4 8000000 - FDG_ISA_RM_WAVES=1 -
