"""GPU dev tool: rate of the library's copy kernel (bench.py's measured_copy) with and without non-temporal accesses."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
for rep in range(2):
    for nt in (0, 1):
        if nt: os.environ.pop("FDG_COPY_PLAIN", None)
        else: os.environ["FDG_COPY_PLAIN"] = "1"
        print("nt" if nt else "plain", "%.0f GB/s" % bench.measured_copy(dev), flush=True)
