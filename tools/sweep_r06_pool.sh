# round 6: what do the pooled kernel's fetches and barriers cost, and does a second wave per SIMD hide them?  (timing experiments; rows with FDG_ISA_DEBUG give garbage results by design)
cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
timeout 1500 python tools/gpu_option_sweep.py gv_ver4_4 524288 - \
  FDG_POOL_WAVES=8 \
  FDG_ISA_DEBUG=nobarrier+nofetchwait \
  FDG_POOL_WAVES=8,FDG_ISA_DEBUG=nobarrier+nofetchwait \
  FDG_POOL_WAVES=8,FDG_ISA_DEBUG=nobarrier+nofetchwait,FDG_COOP_ALIGN=1 \
  FDG_POOL_WAVES=8,FDG_ISA_DEBUG=nobarrier+nofetchwait+nopoolfetch \
  FDG_POOL_FETCH_POLICY=nt "FDG_POOL_FETCH_POLICY=sc1" "FDG_POOL_FETCH_POLICY=sc0 sc1" "FDG_POOL_FETCH_POLICY=sc0 sc1 nt" \
  FDG_POOL_FETCH_WAVES=1 FDG_POOL_FETCH_WAVES=2 \
  FDG_COOP_ALIGN=1 FDG_COOP_ALIGN=1,FDG_ISA_ALIGN=2 \
  FDG_POOL_EPOCH_OPS=256 \
  - 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_log_pool_sweep.txt
timeout 900 python tools/gpu_option_sweep.py parquet_ver4_4 1048576 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_WAVES=8 \
  FDG_ISA_POOL=1,FDG_POOL_WAVES=8,FDG_ISA_DEBUG=nobarrier+nofetchwait FDG_ISA_POOL=1,FDG_ISA_DEBUG=nobarrier+nofetchwait - 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_pool_sweep.txt
