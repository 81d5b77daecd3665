#!/bin/bash
# run the on-device autotuner for every named workload and bring the choices back (dev tool)
mkdir -p gpurun_out/tuned
for w in ${WL:-sigma2 gv_sigma4 gv_sigma4_taylor2 gv_sigma5 gv_sigma6 gv_sigma5_taylor2 sigma4_standin sigma4_worstcase synthetic_small sigma4_taylor_standin}; do
  echo "== $w"
  timeout 900 python bench.py --workload $w --backend isa-autotune --no-cpu-baseline --no-secondary --no-mc-step --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:18], '%.3e evals/s'%d['value'], '%.0f GB/s (%.1f%%)'%(d['roofline']['achieved'], 100*d['roofline']['frac']), d['kernel_info'])"
done
cp feynmandiagram.jl_amd/kernel_cache/fdg_tuned_*.txt gpurun_out/tuned/
for f in gpurun_out/tuned/*.txt; do echo "$f: $(cat $f)"; done
