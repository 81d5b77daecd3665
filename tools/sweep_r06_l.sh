cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 600 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_l.txt; }
: > gpurun_out/r06_log_sweep_l.txt
export SWEEP_LAYOUT=rm
echo "row-major chunk fetches non-temporal, now that every chunk is fetched once (round 4: -20 %, when chunks were fetched again and again)?" | tee -a gpurun_out/r06_log_sweep_l.txt
run parquet_sigma5 2000000 - FDG_ISA_RM_POLICY=nt -
run gv_sigma4_taylor2 4000000 - FDG_ISA_RM_POLICY=nt -
run parquet_sigma4_insdyn 2000000 - FDG_ISA_RM_POLICY=nt -
run gv_sigma5 2000000 - FDG_ISA_RM_POLICY=nt -
run parquet_sigma4_dyn 4000000 - FDG_ISA_RM_POLICY=nt -
run gv_sigma4 8000000 - FDG_ISA_RM_POLICY=nt -
run parquet_sigma4_taylor2 4000000 - FDG_ISA_RM_POLICY=nt -
