cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_j.txt; }
: > gpurun_out/r06_log_sweep_j.txt
echo "alignment pads in the one-wave kernels: in front of every 8-byte instruction (1, the default) or of vector instructions only (2)?" | tee -a gpurun_out/r06_log_sweep_j.txt
run parquet_sigma5 2000000 - FDG_ISA_ALIGN=2 FDG_ISA_ALIGN=0 - FDG_ISA_ALIGN=2
run parquet_sigma4_insdyn 4000000 - FDG_ISA_ALIGN=2 - FDG_ISA_ALIGN=2
run parquet_ver4_4 1048576 - FDG_ISA_ALIGN=2 - FDG_ISA_ALIGN=2
run gv_sigma5 2000000 - FDG_ISA_ALIGN=2 - FDG_ISA_ALIGN=2
run parquet_sigma4_taylor2 8000000 - FDG_ISA_ALIGN=2 -
run gv_sigma4_taylor2 4000000 - FDG_ISA_ALIGN=2 -
run gv_sigma6 500000 - FDG_ISA_ALIGN=2 -
run gv_ver4_4 524288 FDG_ISA_NO_POOL=1 FDG_ISA_NO_POOL=1,FDG_ISA_ALIGN=2
