#!/bin/bash
# round 5: watts per streamed TB/s as a function of the load width (tools/ubench/stream_power.hip), rocm-smi sampled next to each variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/stream_power.txt; : > $O
smi() { /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d[sorted(d)[0]]
g=lambda key: [v for k,v in c.items() if key in k.lower() and 'max' not in k.lower()]
print(g('power (w)')[0])"; }
for v in 0 1 2 3 4; do
  ( tools/ubench/stream_power.bin 6 $v > gpurun_out/sp_$v.log 2>&1 ) & pid=$!
  sleep 2.5; s=""; while kill -0 $pid 2>/dev/null; do s="$s $(smi)"; sleep 0.2; done
  echo "$(cat gpurun_out/sp_$v.log)   watts: $s" >> $O
  sleep 2
done
cat $O
