cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache; chmod 700 /tmp/sweep_cache
for w in parquet_sigma4_insdyn parquet_sigma5; do
timeout 900 python tools/gpu_option_sweep.py $w 2000000 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_WAVES=8 FDG_ISA_COOP=1 FDG_ISA_COOP=1,FDG_COOP_WAVES=4 - 2>&1 | grep -v amdgpu.ids
done
