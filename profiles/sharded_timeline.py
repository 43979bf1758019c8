#!/usr/bin/env python3
"""Kernel-level timeline of ONE iteration from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV pair.

    python profiles/sharded_timeline.py <label>=<trace dir> [<label>=<trace dir> ...]  > profiles/r06_sharded_timeline.md

An iteration = everything from the start of one k_sweep launch to the start of the next.  The table gives, averaged over the
last `N` complete iterations of the trace (steady state: after the warm-up, the preflight and the prior weakenings), for every
launch of the iteration in order: its start relative to the sweep's start, its duration, and the GAP between the end of the
previous launch (on any queue) and its start — what is neither kernel nor copy.  `period` is the iteration itself.
"""
import csv
import glob
import os
import sys

N = 30


def short(name):
    name = name.replace("void ", "").replace("gbp::", "")
    for cut in ("(", "<"):
        if name.startswith("k_sweep") and cut == "<":
            continue
        i = name.find(cut)
        if i > 0:
            name = name[:i]
    return name.replace("__amd_rocclr_", "rocclr:")


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "q" + r["Queue_Id"], int(r["Grid_Size_X"])))
    for f in glob.glob(os.path.join(d, "*memory_copy_trace.csv")):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r["Direction"].replace("MEMORY_COPY_", ""), "dma", 0))
    rows.sort()
    return rows


def timeline(rows):
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k_sweep")]
    its = []
    for a, b in zip(idx[:-1], idx[1:]):
        its.append(rows[a:b] + [rows[b]])
    its = its[-N:]
    # keep the iterations of the most common shape (same launches in the same order)
    shape = lambda it: tuple((r[2], r[3], r[4]) for r in it[:-1])
    shapes = {}
    for it in its:
        shapes.setdefault(shape(it), []).append(it)
    best = max(shapes.values(), key=len)
    n = len(best)
    out = []
    k = len(best[0]) - 1
    for j in range(k):
        st = sum(it[j][0] - it[0][0] for it in best) / n / 1e3
        du = sum(it[j][1] - it[j][0] for it in best) / n / 1e3
        if j == 0:
            gap = None
        else:
            gap = sum(it[j][0] - max(x[1] for x in it[:j]) for it in best) / n / 1e3
        out.append((best[0][j][2], best[0][j][3], best[0][j][4], st, du, gap))
    period = sum(it[-1][0] - it[0][0] for it in best) / n / 1e3
    tail_gap = sum(it[-1][0] - max(x[1] for x in it[:-1]) for it in best) / n / 1e3
    busy = sum(r[4] for r in out)
    return out, period, tail_gap, n, len(its), busy


def main():
    print("# Kernel-level timeline of one iteration (rocprofv3 --kernel-trace --memory-copy-trace; profiles/sharded_timeline.py)\n")
    print("Times in us, means over the last complete iterations of the same shape; `gap` = idle time on the device between the end of "
          "the latest earlier launch of the iteration and this start (negative: the launches overlap, two queues).\n")
    for arg in sys.argv[1:]:
        label, d = arg.split("=", 1)
        rows = load(d)
        tl, period, tail_gap, n, n_all, busy = timeline(rows)
        print("## %s\n" % label)
        print("%d of the last %d iterations share this shape; period %.2f us; sum of the launch durations %.2f us\n" % (n, n_all, period, busy))
        print("| launch | queue | grid (threads) | start | duration | gap before |")
        print("|---|---|---|---|---|---|")
        for name, q, grid, st, du, gap in tl:
            print("| `%s` | %s | %d | %.2f | %.2f | %s |" % (name, q, grid, st, du, "—" if gap is None else "%.2f" % gap))
        print("| next `k_sweep` | | | %.2f | | %.2f |\n" % (period, tail_gap))


if __name__ == "__main__":
    main()
