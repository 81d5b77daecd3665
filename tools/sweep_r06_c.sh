cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_c.txt; }
: > gpurun_out/r06_log_sweep_c.txt
run gv_ver4_4 524288 - FDG_ISA_NO_POOL=1 FDG_ISA_NO_POOL=1,FDG_ISA_NO_STREAMING=1 FDG_ISA_ROOT_POLICY= FDG_COOP_PRIV_LDS=8 -
run gv_sigma6 500000 - FDG_ISA_NO_STREAMING=1 -
run parquet_ver4_4 1048576 - FDG_ISA_POOL=1 -
timeout 1500 python -m pytest tests/test_tile_major.py tests/test_batch_alloc.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_log_tests_c.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or ver4 or edge or golden" 2>&1 | tail -8 | tee -a gpurun_out/r06_log_tests_c.txt
