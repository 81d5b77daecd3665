cd $GRAFT_REPO_ROOT
export FDG_TUNE_VERBOSE=1
WL="gv_ver4_4 parquet_ver4_4 gv_sigma6 parquet_sigma5 parquet_sigma4_insdyn gv_sigma5 parquet_sigma4_taylor2" bash tools/autotune_all.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_log_autotune.txt
