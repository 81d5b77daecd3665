#!/bin/bash
# round 5: watts of the bare fp64 loop with constant and with random operands (tools/ubench/gen_valu_energy.py), rocm-smi sampled next to it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/valu_energy.txt; : > $O
smi() { /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d[sorted(d)[0]]
g=lambda key: [v for k,v in c.items() if key in k.lower() and 'max' not in k.lower()]
print(g('power (w)')[0]+'@'+g('sclk clock speed')[0].strip('()'))"; }
for v in 0 1 0 1; do
  ( tools/ubench/valu_energy.bin 6 $v > gpurun_out/ve_$v.log 2>&1 ) & pid=$!
  sleep 2.5; s=""; while kill -0 $pid 2>/dev/null; do s="$s $(smi)"; sleep 0.2; done
  echo "$(cat gpurun_out/ve_$v.log)   watts@sclk: $s" >> $O
  sleep 2
done
cat $O
