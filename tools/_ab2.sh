export SWEEP_N=15
A="n_reg=120,n_lds=40"; B="n_reg=120,n_lds=80,n_acc=124"
python tools/gpu_cfg_sweep.py gv_sigma6 1000000 - "$A,lookahead_leaf=300,lookahead_mem=128,vn_window=60" "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=400" "$B,lookahead_leaf=100,lookahead_mem=64,vn_window=200" "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=200" "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=1000" 2>&1 | grep -v "Warn\|amdgpu"
python tools/gpu_cfg_sweep.py gv_ver4_4 1000000 - "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=400" "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=200" "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=1000" "$A,lookahead_leaf=300,lookahead_mem=128,vn_window=60" 2>&1 | grep -v "Warn\|amdgpu"
python tools/gpu_cfg_sweep.py sigma4_standin 2000000 - "$B,lookahead_leaf=300,lookahead_mem=128,vn_window=400" "$B,lookahead_leaf=100,lookahead_mem=64,vn_window=200" 2>&1 | grep -v "Warn\|amdgpu"
