cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_f.txt; }
: > gpurun_out/r06_log_sweep_f.txt
echo "alignment pads inside the barrier-synchronised kernels, with the printer's offsets right (round 5: -12 %, because a 0xffffffff was counted as a literal):" | tee -a gpurun_out/r06_log_sweep_f.txt
run gv_ver4_4 524288 - FDG_COOP_ALIGN=1 FDG_COOP_ALIGN=1,FDG_ISA_ALIGN=2 FDG_COOP_ALIGN=1,FDG_POOL_WAVES=8 -
run parquet_ver4_4 1048576 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_COOP_ALIGN=1 -
run sigma4_standin 2000000 - FDG_COOP_ALIGN=1 -
run parquet_ver4_3 4000000 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_COOP_ALIGN=1 -
echo "leaves loaded once in tile-major programs (an evicted leaf goes to LDS / AGPR / the panel instead of being fetched again):" | tee -a gpurun_out/r06_log_sweep_f.txt
run gv_sigma6 500000 - FDG_LEAVES_ONCE=1 -
run gv_sigma5 2000000 - FDG_LEAVES_ONCE=1 -
run gv_ver4_4 524288 FDG_ISA_NO_POOL=1 FDG_ISA_NO_POOL=1,FDG_LEAVES_ONCE=1
run parquet_ver4_4 1048576 - FDG_LEAVES_ONCE=1 -
run parquet_sigma4_taylor2 8000000 - FDG_LEAVES_ONCE=1 -
run sigma4_standin 2000000 FDG_ISA_COOP=0 FDG_ISA_COOP=0,FDG_LEAVES_ONCE=1
( time timeout 900 python -m pytest tests/test_typed.py tests/test_host_api.py -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee -a gpurun_out/r06_log_sweep_f.txt
