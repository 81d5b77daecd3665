# round 6: is the power roof falsifiable?  The same three workloads at the package's default cap and at a lower one:
# the rates should follow (cap - 240 W) / (134 pJ x bytes + 25.5 pJ x fold steps) if the cap is what binds them.
cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
LOG=gpurun_out/r06_log_power_cap_sweep.txt
: > $LOG
smi=/opt/rocm/bin/rocm-smi
$smi --showmaxpower --showpower 2>&1 | grep -v "^$" | tee -a $LOG
probe() { timeout 900 python tools/gpu_power_probe.py 4 "parquet_sigma4 16000000" "parquet_sigma4_taylor2 8000000" "gv_sigma5 2000000" "sigma2 64000000" 2>&1 | grep -v amdgpu.ids | tee -a $LOG; }
echo "== default cap" | tee -a $LOG
probe
for cap in 1100 900; do
  echo "== trying cap $cap W" | tee -a $LOG
  $smi --autorespond y --setpoweroverdrive $cap 2>&1 | grep -v "^$" | tail -4 | tee -a $LOG
  $smi --showmaxpower 2>&1 | grep -i "power" | tee -a $LOG
  probe
done
$smi --autorespond y --resetpoweroverdrive 2>&1 | tail -2 | tee -a $LOG
timeout 600 python -m pytest tests/test_tile_major.py -x -q -m gpu -k repack 2>&1 | tail -3
