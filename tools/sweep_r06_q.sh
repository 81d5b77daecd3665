cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_q.txt; }
: > gpurun_out/r06_log_sweep_q.txt
echo "root stores: scope bits instead of / next to nt (plain allocations)" | tee -a gpurun_out/r06_log_sweep_q.txt
run parquet_sigma4 16000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" "FDG_ISA_ROOT_POLICY=sc1" "FDG_ISA_ROOT_POLICY=sc0" "FDG_ISA_ROOT_POLICY=plain" - "FDG_ISA_ROOT_POLICY=sc0 sc1"
run parquet_sigma4 100000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" - "FDG_ISA_ROOT_POLICY=sc0 sc1"
run gv_sigma4 8000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" -
run sigma2 64000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" -
run parquet_sigma4_dyn 8000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" -
run parquet_sigma4_taylor2 8000000 - "FDG_ISA_ROOT_POLICY=sc0 sc1" -
echo "the headline batch, paired:" | tee -a gpurun_out/r06_log_sweep_q.txt
timeout 900 python tools/gpu_root_policy_paired.py parquet_sigma4 100000000 - "sc0 sc1" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_q.txt
