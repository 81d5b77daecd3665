"""dev: row-major evaluation of a huge graph a few times (run under rocprofv3 --kernel-trace --stats to see the transposition next to the evaluator)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi
name = sys.argv[1] if len(sys.argv) > 1 else "parquet_ver4_4"
t = workloads.get(name); L, R = t.n_leaf, t.n_root
B = 203200
dev = torch.device("cuda:0")
f = fd.compile_table(t, specialize="isa")
rm = torch.empty((B, L), dtype=torch.float64, device=dev)
capi.fill_uniform_device(rm.data_ptr(), B, L, L, 1, 11, 0, torch.cuda.current_stream().cuda_stream)
rr = torch.empty((B, R), dtype=torch.float64, device=dev)
for _ in range(12): f(rr, rm)
torch.cuda.synchronize()
