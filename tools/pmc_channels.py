#!/usr/bin/env python
"""rocprofv3 --pmc ... --output-format json: per-instance (per L2 channel) counter values of every fdg_isa_eval* dispatch.
Prints, per dispatch and counter: duration, sum over instances, max / mean over instances (the imbalance), and the per-XCC sums."""
import collections, glob, json, os, sys
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "**", "*.json"), recursive=True):
    d = json.load(open(f))
    tool = d["rocprofiler-sdk-tool"]
    tool = tool[0] if isinstance(tool, list) else tool
    counters = {}
    for c in tool.get("counters", []):
        counters[c["id"]["handle"] if isinstance(c.get("id"), dict) else c.get("id")] = c
    kern = {}
    for k in tool.get("kernel_symbols", []):
        kern[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name")
    dur = {}
    for r in tool.get("buffer_records", {}).get("kernel_dispatch", []):
        di = r.get("dispatch_info", {})
        dur[di.get("dispatch_id")] = (r.get("end_timestamp", 0) - r.get("start_timestamp", 0)) / 1e6
    recs = tool.get("callback_records", {}).get("counter_collection", [])
    for rec in recs:
        dd = rec.get("dispatch_data", {})
        di = dd.get("dispatch_info", {})
        name = kern.get(di.get("kernel_id"), "?")
        if "fdg_isa_eval" not in str(name): continue
        per = collections.defaultdict(list)
        for x in rec.get("records", []):
            cid = x.get("counter_id"); cid = cid.get("handle") if isinstance(cid, dict) else cid
            per[cid].append((x.get("dimensions") or x.get("dims") or x.get("instance_id") or len(per[cid]), x.get("value", x.get("counter_value"))))
        print(f"dispatch {di.get('dispatch_id')} {str(name)[:40]} ms={dur.get(di.get('dispatch_id'), float('nan')):.3f}")
        for cid, vals in per.items():
            v = [float(b) for _, b in vals]
            nm = counters.get(cid, {}).get("name", cid)
            mean = sum(v) / len(v) if v else 0
            srt = sorted(v)
            print(f"   {nm:38s} n={len(v):4d} sum={sum(v):.6g} mean={mean:.6g} max/mean={max(v) / mean if mean else 0:.3f} min/mean={min(v) / mean if mean else 0:.3f} "
                  f"p90/mean={srt[int(0.9 * (len(v) - 1))] / mean if mean else 0:.3f}")
            if os.environ.get("PMC_FULL"):
                print("      " + " ".join(f"{b:.4g}" for b in v))
