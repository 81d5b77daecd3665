"""GPU dev tool (round 5): socket power and clocks (rocm-smi, sampled twice a second from a side thread) while one workload's evaluation runs back to back for a few
seconds.  Question: are the rows whose two roofline fractions add up to ~1.0-1.2 sitting on the chip's power cap?
usage: gpu_power_probe.py seconds "workload B" ..."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import workloads, capi

dev = torch.device("cuda:0")
secs = float(sys.argv[1])
SMI = "/opt/rocm/bin/rocm-smi"


def sample():
    try:
        out = subprocess.run([SMI, "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        return c
    except Exception as e:
        return {"error": str(e)}


first = sample()
print("idle sample:", {k: v for k, v in first.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk"))}, flush=True)
for spec in sys.argv[2:]:
    name, B = spec.split(); B = int(B)
    t = workloads.get(name); L, R = t.n_leaf, t.n_root
    T = (B + 63) // 64
    st = torch.cuda.current_stream().cuda_stream
    f = fd.compile_table(t, specialize="isa")
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=dev)
    capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    root = torch.zeros((T, R, 64), dtype=torch.float64, device=dev)
    stop, samples = [False], []

    def poll():
        while not stop[0]:
            samples.append(sample()); time.sleep(0.15)
    th = threading.Thread(target=poll); th.start()
    for _ in range(20): f.eval_tiled(root, leaf, B)
    torch.cuda.synchronize()
    n = 0; t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): f.eval_tiled(root, leaf, B)
        torch.cuda.synchronize(); n += 50
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    stop[0] = True; th.join()
    ki = f.kernel_info()
    st_ = t.stats() if hasattr(t, "stats") else {}
    def num(s, key):
        for k, v in s.items():
            if key in k.lower():
                try: return float(str(v).strip("()MmHhzWw ").split()[0].replace("Mhz", "").replace("MHz", ""))
                except Exception: pass
        return float("nan")
    mid = samples[len(samples) // 3:]
    pw = [num(s, "power (w)") if "error" not in s else float("nan") for s in mid]
    keys = sorted({k for s in mid for k in s})
    print(f"{name:26s} {ki['last_kernel']:18s} {ms:7.3f} ms {B / ms / 1e3:8.1f} Mevals/s frac_hbm {8 * (L + R) * B / ms / 1e6 / 8000:.3f} | {len(mid)} samples", flush=True)
    if mid:
        import statistics as S
        def col(key):
            v = []
            for smp in mid:
                for k, x in smp.items():
                    if key in k.lower() and "max" not in k.lower():
                        try: v.append(float(str(x).strip("()").lower().replace("mhz", "")))
                        except Exception: pass
            return v
        for key, unit in (("power (w)", "W"), ("sclk clock speed", "MHz"), ("mclk clock speed", "MHz"), ("fclk clock speed", "MHz")):
            v = col(key)
            if v: print(f"      {key:18s} mean {S.mean(v):7.0f} min {min(v):6.0f} max {max(v):6.0f} {unit}   {[int(x) for x in v]}", flush=True)
    del leaf, root, f
    torch.cuda.empty_cache(); time.sleep(3)
