cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_i.txt; }
: > gpurun_out/r06_log_sweep_i.txt
export SWEEP_LAYOUT=lm
echo "leaf-major at the headline's 1e8 samples (plain allocation): tile order of the persistent waves (a wave takes 2^c consecutive tiles / XCD-contiguous ranges)" | tee -a gpurun_out/r06_log_sweep_i.txt
run parquet_sigma4 100000000 - FDG_ISA_DEBUG=chunk1 FDG_ISA_DEBUG=chunk2 FDG_ISA_DEBUG=chunk3 FDG_ISA_DEBUG=chunk5 FDG_ISA_DEBUG=xcd -
