"""GPU dev tool: how long the final observable sum of bench.py takes for a Julia-layout root matrix."""
import torch, time
dev = torch.device("cuda:0")
for R, B in ((4, 100_000_000), (6, 4_000_000)):
    root = torch.rand((R, B), dtype=torch.float64, device=dev).t()
    for name, fn in (("root.sum(dim=0)", lambda: root.sum(dim=0)), ("root.t().sum(dim=1)", lambda: root.t().sum(dim=1)),
                     ("ones @", lambda: torch.mv(root.t(), torch.ones(B, dtype=torch.float64, device=dev)))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): a = fn()
        torch.cuda.synchronize()
        print(R, B, name, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), a[:2].tolist(), flush=True)
