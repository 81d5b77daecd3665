#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite result: per-kernel count / avg /
min / max / total duration -- the `--kernel-trace --stats` table as text.
usage: rocpd_stats.py [--last=K] results.db [more.db ...]"""
import sqlite3
import sys


def summarise(path, last=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "sum(d.end-d.start), max(d.workgroup_size_x), max(d.grid_size_x), max(d.private_segment_size), "
         "max(d.group_segment_size) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
         "on d.kernel_id = s.id group by s.kernel_name order by 6 desc")
    rows = list(cur.execute(q))
    tot = sum(r[5] for r in rows) or 1
    out = [f"# {path}", f"{'kernel':60s} {'calls':>6s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'total_ms':>10s} {'%':>6s} {'wg':>5s} {'grid':>10s} {'scratch':>8s} {'lds':>7s}"]
    for r in rows:
        out.append(f"{r[0][:60]:60s} {r[1]:6d} {r[2]/1e3:12.2f} {r[3]/1e3:12.2f} {r[4]/1e3:12.2f} {r[5]/1e6:10.3f} {100*r[5]/tot:6.2f} {r[6]:5d} {r[7]:10d} {r[8]:8d} {r[9]:7d}")
    # the timed region of bench.py is its last K launches of the evaluator (after W warm-up launches during which
    # the clocks settle): report that subset too, so it can be compared with bench.py's avg_kernel_ms
    if last:
        for (name,) in list(cur.execute("select distinct s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                                        "on d.kernel_id = s.id where s.kernel_name like 'fdg_isa_eval%' or s.kernel_name like 'fdg_spec%' "
                                        "or s.kernel_name like '%fdg_interp%'")):
            d = [r[0] for r in cur.execute("select d.end-d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                                           "where s.kernel_name = ? order by d.start", (name,))]
            tail = d[-last:]
            if tail:
                out.append(f"{name[:60]:60s} last {len(tail)} launches (bench.py's timed steps and the dozen launches of the clock-probe leg after them: same kernel, same batch): avg_us {sum(tail)/len(tail)/1e3:.2f}  "
                           f"min_us {min(tail)/1e3:.2f}  max_us {max(tail)/1e3:.2f}")
    return "\n".join(out)


if __name__ == "__main__":
    args = sys.argv[1:]
    last = 0
    if args and args[0].startswith("--last="):
        last = int(args[0].split("=")[1]); args = args[1:]
    for p in args:
        print(summarise(p, last))
        print()
