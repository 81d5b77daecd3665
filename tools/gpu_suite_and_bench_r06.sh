# round 6: the whole GPU suite, then the default bench line (what the driver runs at round end)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r06_log_gpu_suite.txt 2>&1
( time timeout 900 python bench.py > gpurun_out/r06_a_bench_line.json 2> gpurun_out/r06_a_bench_stderr.txt ) 2>> gpurun_out/r06_log_gpu_suite.txt
cp bench_detail.json gpurun_out/r06_a_bench_detail.json 2>/dev/null
tail -c 3000 gpurun_out/r06_a_bench_stderr.txt | grep -v "^\[bench\] detail" | tail -5
cat gpurun_out/r06_log_gpu_suite.txt
