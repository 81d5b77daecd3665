cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_h.txt; }
: > gpurun_out/r06_log_sweep_h.txt
export SWEEP_LAYOUT=rm
echo "row-major, two waves per SIMD with leaves loaded once (panel accesses allowed up to PCT % of the fold steps):" | tee -a gpurun_out/r06_log_sweep_h.txt
run gv_sigma5 2000000 - FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=3 FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=3,FDG_RM_VN=100 -
run parquet_sigma5 2000000 - FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=5 FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=8,FDG_RM_VN=1000 -
run parquet_sigma4_insdyn 2000000 - FDG_ISA_RM_WAVES=2,FDG_ISA_RM_PANEL_PCT=8 -
run gv_sigma4 8000000 -
run parquet_sigma4_taylor2 4000000 -
