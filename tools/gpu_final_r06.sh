cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-c}
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_log_gpu_suite_$TAG.txt 2>&1
( time python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -12 ) >> gpurun_out/r06_log_gpu_suite_$TAG.txt 2>&1
( time timeout 900 python bench.py > gpurun_out/r06_${TAG}_bench_line.json 2> gpurun_out/r06_${TAG}_bench_stderr.txt ) 2>> gpurun_out/r06_log_gpu_suite_$TAG.txt
cp bench_detail.json gpurun_out/r06_${TAG}_bench_detail.json 2>/dev/null
cat gpurun_out/r06_log_gpu_suite_$TAG.txt | grep -v Warning | tail -30
