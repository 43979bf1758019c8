#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary."""
import csv
import glob
import os
import sys

out_dir, dst = sys.argv[1], sys.argv[2]
lines = ["# rocprofv3 summary (%s)" % os.path.basename(out_dir), ""]


def find(pattern):
    return sorted(glob.glob(os.path.join(out_dir, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    lines += ["## kernel stats (%s)" % os.path.relpath(f, out_dir), "", "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        lines.append("| %s | %s | %s | %s | %s |" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                                   r.get("AverageNs"), r.get("Percentage")))
    lines.append("")

for tag in ("fetch", "write"):
    for f in find("*%s*counter_collection.csv" % tag):
        agg = {}
        if "pmc" not in f and "fetch" not in os.path.basename(f) and "write" not in os.path.basename(f):
            continue
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:60], r.get("Counter_Name"))
            v = float(r.get("Counter_Value", 0))
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += v
        lines += ["## PMC %s (%s)" % (tag, os.path.relpath(f, out_dir)), "", "| kernel | counter | dispatches | mean value per dispatch |", "|---|---|---|---|"]
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append("| %s | %s | %d | %.1f |" % (k, c, n, s / n))
        lines.append("")
# HBM traffic per launch as bench.py derived it from these passes (read side of streaming kernels doubled: gfx950 correction)
for f in find("traffic_S1.json"):
    lines += ["## HBM traffic per launch (%s)" % os.path.relpath(f, out_dir), "", "```", open(f).read().strip(), "```", ""]
for f in find("*bench.json"):
    lines += ["## %s" % os.path.basename(f), "", "```", open(f).read().strip()[-3000:], "```", ""]
open(dst, "w").write("\n".join(lines))
print("\n".join(lines[:60]))
