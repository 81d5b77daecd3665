cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in f g h; do
  timeout 900 python bench.py > gpurun_out/r06_${t}_bench_line.json 2> /dev/null
  python - <<PY
import json
l=json.loads(open('gpurun_out/r06_${t}_bench_line.json').read()); r=l['roofline']
print('${t}', l['value'], r['frac'], r.get('frac_from_ms_per_step'), r.get('frac_hbm_min_over_steps'), l['config']['device'], l['config5']['value'], l['accumulate']['frac_hbm'])
PY
done
