#!/bin/bash
for G in gv_sigma5 sigma4_standin sigma4_worstcase gv_sigma6 synthetic_small; do
for opt in "n_reg=120,n_lds=40,lookahead_leaf=300" "n_reg=120,n_lds=80,n_acc=124,lookahead_leaf=300" "n_reg=120,n_lds=80,lookahead_leaf=300"; do
  echo "== $G $opt"; python tools/gpu_isa_check.py $G --timeonly --opt=$opt 2>&1 | grep TIME
done
done
python tools/gpu_isa_check.py sigma4_standin,synthetic_small --opt=n_reg=120,n_lds=80,n_acc=124 2>&1 | grep -v "exact\|amdgpu"
