cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_d.txt; }
: > gpurun_out/r06_log_sweep_d.txt
run gv_ver4_4 524288 - FDG_POOL_VADDR=1 FDG_ISA_NO_POOL=1 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=-1 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=16 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=0 FDG_ISA_NO_POOL=1,FDG_ISA_NT_DIST=128 -
run gv_sigma6 500000 - FDG_ISA_NT_DIST=-1 FDG_ISA_NT_DIST=0 FDG_ISA_NT_DIST=16 FDG_ISA_NT_DIST=128 -
run parquet_ver4_4 1048576 - FDG_ISA_NT_DIST=-1 FDG_ISA_NT_DIST=16 FDG_ISA_NT_DIST=128 FDG_ISA_NT_DIST=0 -
run gv_sigma5 2000000 - FDG_ISA_NT_DIST=-1 FDG_ISA_NT_DIST=16 FDG_ISA_NT_DIST=128 -
run parquet_sigma4_taylor2 8000000 - FDG_ISA_NT_DIST=-1 -
export SWEEP_LAYOUT=rm
echo "row-major:" | tee -a gpurun_out/r06_log_sweep_d.txt
run parquet_sigma5 2000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 FDG_RM_LEAVES_ONCE=1,FDG_ISA_RM_BUFS=4 FDG_RM_LEAVES_ONCE=1,FDG_ISA_RM_BUFS=3 FDG_RM_LEAVES_ONCE=0,FDG_ISA_RM_BUFS=2 -
run gv_sigma4_taylor2 4000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 -
run parquet_sigma4_insdyn 2000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 -
run gv_sigma5 2000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 -
run parquet_sigma4_taylor2 4000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 -
run parquet_sigma4_dyn 4000000 - FDG_RM_LEAVES_ONCE=0 FDG_RM_LEAVES_ONCE=1 -
