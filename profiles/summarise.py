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
# HBM traffic of the dominant kernel per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE tallies wide coalesced streaming reads at 1/2.
import json
traffic = {}
for tag in ("fetch", "write"):
    for f in find("*%s*counter_collection.csv" % tag):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if r.get("Kernel_Name", "").startswith("void gbp::k_sweep") or "k_sweep" in r.get("Kernel_Name", ""):
                tot += float(r.get("Counter_Value", 0)); n += 1
        if n:
            traffic[tag + "_kb_per_launch"] = tot / n
if len(traffic) == 2:
    traffic["hbm_bytes_per_launch"] = int((2.0 * traffic["fetch_kb_per_launch"] + traffic["write_kb_per_launch"]) * 1024)
    traffic["kernel"] = "k_sweep"
    traffic["workload"] = "S1 1000x100000x1000000"
    traffic["note"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; read side doubled (gfx950 correction)"
    json.dump(traffic, open(os.path.join(os.path.dirname(dst), "traffic_S1.json"), "w"), indent=1)
    lines += ["## k_sweep HBM traffic per launch", "", "```", json.dumps(traffic, indent=1), "```", ""]
for f in find("*_bench.json"):
    lines += ["## %s" % os.path.basename(f), "", "```", open(f).read().strip()[-3000:], "```", ""]
open(dst, "w").write("\n".join(lines))
print("\n".join(lines[:60]))
