#!/bin/bash
# round 5: socket power (rocm-smi) under the two pure loads -- streaming (the library's copy kernel through tools/gpu_copy_probe.py) and fp64 arithmetic
# (tools/ubench/valu_align.bin, aligned body, one and two waves per SIMD) -- to price a byte and an operation for the power-cap roof of DESIGN 6b
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/power_calib.txt; : > $O
smi() { /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d[sorted(d)[0]]
g=lambda key: [v for k,v in c.items() if key in k.lower() and 'max' not in k.lower()]
print(g('power (w)')[0], g('sclk clock speed')[0].strip('()'))"; }
watch() { tag=$1; shift; ( "$@" > gpurun_out/calib_$tag.log 2>&1 ) & pid=$!; sleep 2; s=""; while kill -0 $pid 2>/dev/null; do s="$s $(smi | tr ' ' '@')"; sleep 0.2; done; echo "$tag: $s" >> $O; tail -3 gpurun_out/calib_$tag.log | cut -c1-200 >> $O; }
echo "idle: $(smi) $(smi)" >> $O
watch valu bash -c 'for i in $(seq 1 60); do tools/ubench/valu_align.bin | grep "P0"; done'
sleep 3
cat > /tmp/fdg_copy_loop.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from feynmandiagram_jl_amd import capi
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
t0 = time.time(); k = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 8:
    for _ in range(100): capi.copy_device(b.data_ptr(), a.data_ptr(), n // 8, st)
    torch.cuda.synchronize(); k += 100
e1.record(); torch.cuda.synchronize()
print("copy", 2 * n * k / e0.elapsed_time(e1) / 1e9, "TB/s (read + write)")
PY
watch copy python /tmp/fdg_copy_loop.py
cat $O
