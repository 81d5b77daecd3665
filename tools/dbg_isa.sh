#!/bin/bash
for G in gv_sigma5 gv_sigma4_taylor2 gv_sigma6; do
for w in 1 200 1000 5000 0; do
 for cfg in "n_reg=120,n_lds=40,lookahead_leaf=300" "n_reg=120,n_lds=80,n_acc=124,lookahead_leaf=100,lookahead_mem=64"; do
  echo "== $G vn=$w $cfg"; python tools/gpu_isa_check.py $G --timeonly --opt=$cfg,vn_window=$w 2>&1 | grep TIME
 done
done
done
