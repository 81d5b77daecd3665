# round 6: the whole GPU suite, then the default bench line (what the driver runs at round end)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/sweep_cache; chmod 700 /tmp/sweep_cache
TAG=${1:-b}
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r06_log_gpu_suite.txt 2>&1
cat gpurun_out/r06_log_gpu_suite.txt
( time timeout 900 python bench.py > gpurun_out/r06_${TAG}_bench_line.json 2> gpurun_out/r06_${TAG}_bench_stderr.txt ) 2>> gpurun_out/r06_log_gpu_suite.txt
cp bench_detail.json gpurun_out/r06_${TAG}_bench_detail.json 2>/dev/null
grep -v "^\[bench\] detail" gpurun_out/r06_${TAG}_bench_stderr.txt | tail -5
tail -8 gpurun_out/r06_log_gpu_suite.txt
export SWEEP_LAYOUT=rm
for w in "gv_ver4_4 524288" "parquet_ver4_4 1048576" "gv_sigma6 500000"; do set -- $w; timeout 600 python tools/gpu_option_sweep.py $1 $2 - FDG_ISA_NO_RM=1 - 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_rm_big.txt; done
