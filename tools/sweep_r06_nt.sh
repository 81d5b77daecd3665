# round 6: non-temporal leaf loads everywhere (not only a leaf's last load of the tile)?  One-wave kernels, tile-major batches.
cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 900 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_nt_sweep.txt; }
: > gpurun_out/r06_log_nt_sweep.txt
run gv_sigma6 500000 - FDG_ISA_LEAF_POLICY=nt "FDG_ISA_LEAF_POLICY=sc1" -
run parquet_ver4_4 1048576 - FDG_ISA_LEAF_POLICY=nt FDG_ISA_LEAF_POLICY=nt,FDG_ISA_PANEL_POLICY=nt FDG_ISA_PANEL_POLICY=nt -
run parquet_sigma4_insdyn 4000000 - FDG_ISA_LEAF_POLICY=nt -
run parquet_sigma5 2000000 - FDG_ISA_LEAF_POLICY=nt -
run gv_sigma5 2000000 - FDG_ISA_LEAF_POLICY=nt -
run parquet_sigma4_taylor2 8000000 - FDG_ISA_LEAF_POLICY=nt -
run gv_sigma4_taylor2 4000000 - FDG_ISA_LEAF_POLICY=nt -
run sigma4_standin 2000000 - FDG_ISA_LEAF_POLICY=nt FDG_ISA_COOP=0 FDG_ISA_COOP=0,FDG_ISA_LEAF_POLICY=nt FDG_ISA_COOP=0,FDG_ISA_LEAF_POLICY=nt,FDG_ISA_PANEL_POLICY=nt -
run parquet_sigma4_dyn 8000000 - FDG_ISA_LEAF_POLICY=nt -
run parquet_sigma4 16000000 - FDG_ISA_LEAF_POLICY=nt -
