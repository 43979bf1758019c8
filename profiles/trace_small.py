#!/usr/bin/env python3
"""Per-launch kernel durations of a small-sequence CLI run, bucketed by iteration: how the sweep's time moves once
relinearisations set in.  Input: the kernel-trace CSV of
    rocprofv3 --kernel-trace --output-format csv -d DIR -o small -- gbp_poplar_amd/bin/ba --bal_file ... --eval_every 100"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sw = [(s, e) for s, e, n in rows if "k_sweep" in n or "k_persist" in n]
be = [(s, e) for s, e, n in rows if "k_beliefs" in n]
print("| launches | k_sweep avg us | min | max | gap to next kernel avg us |")
print("|---|---|---|---|---|")
starts = sorted(s for s, e, n in rows)
import bisect
for lo, hi in ((0, 10), (10, 25), (25, 60), (60, 150), (150, 400), (400, 100000)):
    part = sw[lo:hi]
    if not part:
        continue
    dur = [(e - s) / 1e3 for s, e in part]
    gaps = []
    for s, e in part:
        i = bisect.bisect_right(starts, s)
        if i < len(starts):
            gaps.append((starts[i] - e) / 1e3)
    print("| %d-%d | %.2f | %.2f | %.2f | %.2f |" % (lo, min(hi, len(sw)), sum(dur) / len(dur), min(dur), max(dur), sum(gaps) / max(len(gaps), 1)))
if be:
    dur = [(e - s) / 1e3 for s, e in be]
    print("\nk_beliefs: %d launches, avg %.2f us, min %.2f, max %.2f" % (len(dur), sum(dur) / len(dur), min(dur), max(dur)))
if rows:
    print("span first..last kernel: %.3f ms for %d kernels" % ((rows[-1][1] - rows[0][0]) / 1e6, len(rows)))
