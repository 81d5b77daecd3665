#!/bin/bash
for w in 1 2 3 4; do
  echo "== sigma4_standin waves/CU=$w"; FDG_ISA_WAVES_PER_CU=$w python tools/gpu_isa_check.py sigma4_standin --timeonly 2>&1 | grep TIME
done
for opt in "n_reg=120,n_lds=100,n_acc=124" "n_reg=120,n_lds=80,n_acc=124,lookahead_mem=64" "n_reg=120,n_lds=80,n_acc=124,lookahead_mem=300" "n_reg=120,n_lds=80,n_acc=124,lookahead_leaf=100"; do
  echo "== sigma4_standin $opt"; python tools/gpu_isa_check.py sigma4_standin --timeonly --opt=$opt 2>&1 | grep TIME
done
