cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 600 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_m.txt; }
: > gpurun_out/r06_log_sweep_m.txt
export SWEEP_LAYOUT=rm
echo "what the chunked row-major kernel's time is made of (timing experiments: garbage results by design)" | tee -a gpurun_out/r06_log_sweep_m.txt
run parquet_sigma5 2000000 - FDG_ISA_DEBUG=novalu FDG_ISA_DEBUG=noleaf FDG_ISA_DEBUG=noleaf+novalu FDG_ISA_DEBUG=nolds+noacc FDG_ISA_DEBUG=novmwait
run gv_sigma4_taylor2 4000000 - FDG_ISA_DEBUG=novalu FDG_ISA_DEBUG=noleaf FDG_ISA_DEBUG=noleaf+novalu FDG_ISA_DEBUG=nolds+noacc FDG_ISA_DEBUG=novmwait
unset SWEEP_LAYOUT
echo "tile-major, the same graphs:" | tee -a gpurun_out/r06_log_sweep_m.txt
run parquet_sigma5 2000000 - FDG_ISA_DEBUG=novalu FDG_ISA_DEBUG=noleaf
run gv_sigma4_taylor2 4000000 - FDG_ISA_DEBUG=novalu FDG_ISA_DEBUG=noleaf
