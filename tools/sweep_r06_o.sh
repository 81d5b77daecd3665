cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 600 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_o.txt; }
: > gpurun_out/r06_log_sweep_o.txt
echo "pooled kernel: the waves' value-numbering window, pool read-ahead, private LDS slots" | tee -a gpurun_out/r06_log_sweep_o.txt
run gv_ver4_4 524288 - FDG_POOL_VN=200 FDG_POOL_VN=600 FDG_POOL_VN=1000 FDG_POOL_VN=100 FDG_POOL_READ_AHEAD=160 FDG_POOL_LA_LDS=64 FDG_COOP_PRIV_LDS=12 FDG_POOL_LEAF_COST=1.0 -
