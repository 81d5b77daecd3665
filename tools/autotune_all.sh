#!/bin/bash
# run the on-device autotuner for every named workload and bring the choices back (dev tool): the tuned-parameter files
# land in gpurun_out/tuned_new/ (copy the ones worth keeping into feynmandiagram.jl_amd/kernel_cache/)
mkdir -p gpurun_out/tuned_new
export FDG_CACHE_DIR=$PWD/gpurun_out/tuned_new FDG_IGNORE_TUNED=1
chmod 700 gpurun_out/tuned_new
for w in ${WL:-sigma2 gv_sigma4 gv_sigma4_taylor2 gv_sigma5 gv_sigma6 gv_sigma5_taylor2 sigma4_standin sigma4_worstcase synthetic_small sigma4_taylor_standin}; do
  echo "== $w"
  before=$(ls gpurun_out/tuned_new/fdg_tuned_*.txt 2>/dev/null | sort)
  timeout 900 python bench.py --workload $w --backend isa-autotune --no-cpu-baseline --no-secondary --no-mc-step --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:24], '%.3e evals/s'%d['value'], r['bound'], 'frac %.3f hbm %.3f valu %s'%(r['frac'], r['frac_hbm'], r['frac_valu']), r['kernel'])"
  after=$(ls gpurun_out/tuned_new/fdg_tuned_*.txt 2>/dev/null | sort)
  for f in $(comm -13 <(echo "$before") <(echo "$after")); do echo "   $w -> $(basename $f): $(cat $f)"; done
done
rm -f gpurun_out/tuned_new/*.hsaco
