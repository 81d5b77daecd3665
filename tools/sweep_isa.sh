#!/bin/bash
# parameter sweep of the ISA back end on one graph (dev tool)
G=${1:-gv_sigma5}
for opt in "n_reg=120,n_lds=80,lookahead_mem=160,lookahead_lds=24" "n_reg=120,n_lds=80,lookahead_mem=600,lookahead_lds=48" "n_reg=120,n_lds=80,lookahead_mem=2000,lookahead_lds=64" "n_reg=120,n_lds=80,lookahead_mem=6000,lookahead_lds=100" "n_reg=60,n_lds=40,lookahead_mem=600,lookahead_lds=48" "n_reg=60,n_lds=40,lookahead_mem=2000,lookahead_lds=64" "n_reg=40,n_lds=20,lookahead_mem=1000,lookahead_lds=48" "n_reg=28,n_lds=10,lookahead_mem=600,lookahead_lds=32"; do
  echo "== $opt"; python tools/gpu_isa_check.py $G --timeonly --opt=$opt 2>&1 | grep TIME
done
