cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4 5; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-mc-step --no-cpu-baseline > gpurun_out/r05_l_bench_line_$i.json 2>gpurun_out/r05_l_$i.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05_l_bench_line_$i.json")); r=d["roofline"]
print(d["config"].get("device"), d["value"], r["frac"], r["frac_hbm_min_over_steps"], r["placement"], d["accumulate"]["frac_hbm"])
PY
grep -o '"pairs_timed": [0-9]*\|"seconds": [0-9.]*, "seconds_settling": [0-9.]*\|"fillers": [0-9]*\|"mapped_pairs_frac_min": [0-9.]*' gpurun_out/r05_l_$i.err | tr '\n' ' '; echo
done
