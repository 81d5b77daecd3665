cd $GRAFT_REPO_ROOT; mkdir -p /tmp/sweep_cache gpurun_out; chmod 700 /tmp/sweep_cache
run() { w=$1; b=$2; shift 2; timeout 600 python tools/gpu_option_sweep.py $w $b "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_log_sweep_p.txt; }
: > gpurun_out/r06_log_sweep_p.txt
echo "scope bits next to nt on the leaf stream (graphs that load every leaf once: the policy applies to the same loads as the default's)" | tee -a gpurun_out/r06_log_sweep_p.txt
run parquet_sigma5 2000000 - "FDG_ISA_LEAF_POLICY=sc1 nt" "FDG_ISA_LEAF_POLICY=sc0 nt" "FDG_ISA_LEAF_POLICY=sc0 sc1 nt" "FDG_ISA_LEAF_POLICY=sc0 sc1" -
run parquet_sigma4 16000000 - "FDG_ISA_LEAF_POLICY=sc1 nt" "FDG_ISA_LEAF_POLICY=sc0 nt" "FDG_ISA_LEAF_POLICY=sc0 sc1 nt" "FDG_ISA_ROOT_POLICY=sc1 nt" "FDG_ISA_ROOT_POLICY=sc0 sc1 nt" "FDG_ISA_ROOT_POLICY=sc0 sc1" -
run gv_sigma4_taylor2 4000000 - "FDG_ISA_LEAF_POLICY=sc1 nt" "FDG_ISA_LEAF_POLICY=sc0 sc1 nt" -
