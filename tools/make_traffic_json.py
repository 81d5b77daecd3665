#!/usr/bin/env python
"""profiles/r04_traffic.json from the PMC passes of tools/prof_pmc_list.sh: per workload and layout, HBM-side bytes per
evaluation of the evaluator kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / samples  (FETCH_SIZE / WRITE_SIZE in KiB per
dispatch; the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section, calibrated on sigma2 where
the byte count is known: 64 B read + 16 B written per evaluation).  usage: make_traffic_json.py gpurun_out/prof_<tag> out.json [r03]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root, out_path = sys.argv[1], sys.argv[2]
TAG = sys.argv[3] if len(sys.argv) > 3 else "r02"        # prefix of the summaries kept under profiles/
import bench
SAMPLES = bench.DEFAULT_B           # the batch bench.py --workload W runs without --samples
if os.path.exists(out_path):          # keep the entries of earlier profile sets: only the workloads found under `root` are replaced
    prev = json.load(open(out_path))
else:
    prev = {}
out = {"_comment": "HBM-side traffic of the evaluator kernel from separate rocprofv3 --pmc passes (tools/prof_r02.sh; summaries in "
                   "profiles/r02_pmc_*.txt): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; factor 2 = the guide's gfx950 FETCH_SIZE "
                   "correction, calibrated on sigma2 (80 B per evaluation).  Infinity-Cache hits are included in FETCH_SIZE: this is "
                   "L2-miss traffic, an upper bound on HBM bytes.  Keys: workload (leaf-major matrix), workload:tile_major, workload:sample_major."}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[4:]
    lay = "sample_major" if name.endswith("_sample_major") else ("tile_major" if name.endswith("_tile_major") else "leaf_major")
    wl = name[: -len("_" + lay)]
    vals = {}
    for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if k.startswith("fdg_isa_eval") and "_acc" not in k:       # the evaluator, not the accumulate leg bench.py runs after it
                    vals.setdefault((k.replace(".kd", ""), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    kernels = sorted({k for k, _ in vals})
    fetch = sum(sum(vals.get((k, "FETCH_SIZE"), [0])) / max(1, len(vals.get((k, "FETCH_SIZE"), [0]))) for k in kernels)
    write = sum(sum(vals.get((k, "WRITE_SIZE"), [0])) / max(1, len(vals.get((k, "WRITE_SIZE"), [0]))) for k in kernels)
    if not kernels:
        continue
    B = SAMPLES[wl]
    key = wl if lay == "leaf_major" else wl + ":" + lay
    out[key] = {"layout": lay, "samples": B, "kernels": kernels, "fetch_kib": fetch, "write_kib": write,
                "bytes_per_eval": round((2 * fetch + write) * 1024 / B, 1), "source": f"profiles/{TAG}_pmc_{wl}_{lay}.txt"}
for k, v in prev.items():
    out.setdefault(k, v)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
