# round 6, second sweep: the pooled kernel with non-temporal fetches (the default now) -- what is left to barriers, fetch count, epochs, wave count
cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
timeout 1500 python tools/gpu_option_sweep.py gv_ver4_4 524288 - \
  FDG_POOL_FETCH_POLICY=plain \
  FDG_ISA_DEBUG=nobarrier+nofetchwait \
  FDG_ISA_DEBUG=nobarrier+nofetchwait+nopoolfetch \
  FDG_ISA_DEBUG=nopoolfetch \
  FDG_ISA_DEBUG=nobarrier+nofetchwait+norecv+nopoolfetch \
  FDG_POOL_WAVES=8 \
  FDG_POOL_WAVES=8,FDG_ISA_DEBUG=nobarrier+nofetchwait \
  FDG_POOL_EPOCH_OPS=256 FDG_POOL_EPOCH_OPS=192 FDG_POOL_EPOCH_OPS=96 FDG_POOL_EPOCH_OPS=64 \
  FDG_POOL_AHEAD=4 FDG_POOL_AHEAD=12 FDG_POOL_AHEAD=16 \
  FDG_POOL_UNIT=2 FDG_POOL_PAIR=1 FDG_POOL_PAIR=1,FDG_POOL_PAIR_FAR=0 \
  FDG_POOL_READ_AHEAD=64 FDG_POOL_READ_AHEAD=128 \
  FDG_COOP_ALIGN=1 \
  FDG_POOL_FETCH_WAVES=2 \
  FDG_ISA_NO_POOL=1 \
  "FDG_ISA_NO_POOL=1,FDG_ISA_LEAF_POLICY=nt" \
  - 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_log_pool_sweep2.txt
timeout 900 python tools/gpu_option_sweep.py parquet_ver4_4 1048576 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_WAVES=8 \
  FDG_ISA_POOL=1,FDG_ISA_DEBUG=nobarrier+nofetchwait FDG_ISA_POOL=1,FDG_POOL_EPOCH_OPS=256 FDG_ISA_POOL=1,FDG_POOL_EPOCH_OPS=64 - 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_pool_sweep2.txt
for w in gv_sigma6 parquet_ver4_3 sigma4_standin; do
timeout 600 python tools/gpu_option_sweep.py $w 1048576 - FDG_ISA_POOL=1 FDG_ISA_POOL=1,FDG_POOL_FETCH_POLICY=plain 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_pool_sweep2.txt
done
