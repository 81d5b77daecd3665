#!/bin/bash
# round 5: the one-wave kernels keep few loads in flight (their memory stream alone runs at 0.5-0.74 of the roof): how far ahead should leaf loads be hoisted?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/lookahead_sweep.txt; : > $O
export SWEEP_LAYOUT=tile_major
B1="n_reg=120,n_lds=80,n_acc=124,lookahead_lds=32,lookahead_mem=128"
run() { w=$1; b=$2; vn=$3; shift 3; args=""; for la in 100 300 600 1000 1500 2500 4000; do args="$args $B1,lookahead_leaf=$la,vn_window=$vn"; done
  timeout 900 python tools/gpu_cfg_sweep.py $w $b $args 2>&1 | grep -v "Warning\|amdgpu.ids" >> $O; }
run parquet_sigma4_insdyn 4000000 2000
run parquet_sigma5 4000000 0
run parquet_ver4_4 1000000 1000
run parquet_sigma4_dyn 8000000 0
run gv_sigma4_taylor2 8000000 0
cat $O
