cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 300 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_k.txt; }
: > gpurun_out/r06_log_sweep_k.txt
echo "pooled kernel: progress words instead of s_barrier between the epochs of a tile (FDG_POOL_SYNC=flags; slack = epochs a wave may run ahead)" | tee -a gpurun_out/r06_log_sweep_k.txt
run parquet_ver4_3 1048576 FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_SYNC=flags,FDG_POOL_SLACK=1
run gv_ver4_4 524288 - FDG_POOL_SYNC=flags,FDG_POOL_SLACK=1 FDG_POOL_SYNC=flags,FDG_POOL_SLACK=2 FDG_POOL_SYNC=flags,FDG_POOL_SLACK=1,FDG_POOL_EPOCH_OPS=256 FDG_POOL_SYNC=flags,FDG_POOL_SLACK=2,FDG_POOL_EPOCH_OPS=256 FDG_POOL_SYNC=flags,FDG_POOL_SLACK=1,FDG_POOL_WAVES=8 -
run parquet_ver4_4 1048576 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_SYNC=flags,FDG_POOL_SLACK=1 FDG_ISA_POOL=1,FDG_POOL_SYNC=flags,FDG_POOL_SLACK=2
