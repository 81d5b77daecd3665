#!/bin/bash
# GPU dev tool: row-major ([B, L]) rate of the ISA back end's variant per workload under a few settings, next to the leaf-major rate.
# usage: tools/gpu_rm_sweep.sh "SETTING1" "SETTING2" ... -- workload ...      (settings as in gpu_env_sweep.py; "-" = none)
settings=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do settings+=("$1"); shift; done; shift
for w in "$@"; do
  SWEEP_CACHE=/tmp/fdg-sweep-lm python tools/gpu_env_sweep.py $w - 2>&1 | grep -v Warn | sed 's/^/leaf-major  /'
  SWEEP_LAYOUT=sample_major SWEEP_CACHE=/tmp/fdg-sweep-rm python tools/gpu_env_sweep.py $w "${settings[@]}" 2>&1 | grep -v Warn | sed 's/^/row-major   /'
done
