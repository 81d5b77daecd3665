cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_e.txt; }
: > gpurun_out/r06_log_sweep_e.txt
export SWEEP_LAYOUT=rm
echo "row-major (old selection = FDG_RM_LEAVES_ONCE=0 with the tile-major kernel's value-numbering window):" | tee -a gpurun_out/r06_log_sweep_e.txt
run parquet_sigma5 2000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=0 FDG_RM_VN=1000 FDG_RM_VN=400 FDG_RM_VN=2000 FDG_RM_VN=1000,FDG_ISA_RM_BUFS=4 FDG_RM_VN=1000,FDG_ISA_RM_BUFS=3 FDG_RM_VN=1000,FDG_ISA_RM_LA=96 FDG_RM_VN=1000,FDG_ISA_RM_LA=24 -
run gv_sigma4_taylor2 4000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=200 FDG_RM_VN=400 FDG_RM_VN=1000 -
run parquet_sigma4_insdyn 2000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=0 FDG_RM_VN=2000 FDG_RM_VN=1000 -
run gv_sigma5 2000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=200 FDG_RM_VN=400 FDG_RM_VN=1000 -
run parquet_sigma4_taylor2 4000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=200 -
run parquet_sigma4_dyn 4000000 - FDG_RM_LEAVES_ONCE=0,FDG_RM_VN=200 -
run gv_sigma4 8000000 -
unset SWEEP_LAYOUT
echo "tile-major:" | tee -a gpurun_out/r06_log_sweep_e.txt
run gv_ver4_4 524288 - FDG_ISA_NO_POOL=1 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=256 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=512 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=1024 -
run gv_sigma6 500000 - FDG_ISA_NT_DIST=512 -
export SWEEP_LAYOUT=lm
echo "leaf-major:" | tee -a gpurun_out/r06_log_sweep_e.txt
run gv_ver4_4 524288 - FDG_ISA_NO_POOL=1 -
run gv_sigma6 500000 -
