#!/usr/bin/env python3
"""Per-dispatch durations of k_sweep / k_beliefs from a rocprofv3 --kernel-trace CSV, ordinary vs lock-step dispatches.

  python3 profiles/relin_dispatches.py <dir with *kernel_trace.csv> [<fetch pmc dir> <write pmc dir>] > profiles/r04_relin_dispatches.csv

On the converging synthetic graph every factor relinearises in the same sweep once per 11 sweeps (damping_count -8 -> 3,
gbp_codelets.cpp:280): those launches write every potential back and run relin_core on all lanes.  A dispatch is labelled
`lockstep` when its duration exceeds 1.25 x the median of its kernel.  With PMC directories (rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes of the same command: dispatch order is the same) the per-dispatch counters are added.
"""
import csv
import glob
import os
import statistics
import sys


def rows(d, pattern):
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True)):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    trace = rows(sys.argv[1], "*kernel_trace.csv")
    per = {}
    for r in trace:
        name = r["Kernel_Name"]
        short = "k_sweep" if "k_sweep" in name else "k_beliefs" if "k_beliefs" in name else None
        if not short:
            continue
        per.setdefault(short, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    pmc = {}
    for tag, d in zip(("FETCH_SIZE", "WRITE_SIZE"), sys.argv[2:4]):
        for r in rows(d, "*counter_collection.csv"):
            if r.get("Counter_Name") != tag:
                continue
            name = r["Kernel_Name"]
            short = "k_sweep" if "k_sweep" in name else "k_beliefs" if "k_beliefs" in name else None
            if short:
                pmc.setdefault((short, tag), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatch", "duration_us", "kind", "fetch_size_kb", "write_size_kb", "hbm_mb_(2xfetch+write)"])
    for k, v in per.items():
        v.sort()
        med = statistics.median(d for _, d in v)
        f = [x for _, x in sorted(pmc.get((k, "FETCH_SIZE"), []))]
        wr = [x for _, x in sorted(pmc.get((k, "WRITE_SIZE"), []))]
        for i, (_, d) in enumerate(v):
            fk = f[i] if i < len(f) and len(f) == len(v) else ""
            wk = wr[i] if i < len(wr) and len(wr) == len(v) else ""
            mb = round((2 * fk + wk) * 1024 / 1e6, 1) if fk != "" and wk != "" else ""
            w.writerow([k, i, round(d / 1e3, 2), "lockstep" if d > 1.25 * med else "ordinary", fk, wk, mb])
    for k, v in per.items():
        med = statistics.median(d for _, d in v)
        o = [d for _, d in v if d <= 1.25 * med]
        l = [d for _, d in v if d > 1.25 * med]
        sys.stderr.write("%s: %d dispatches, mean %.2f us; ordinary %d mean %.2f min %.2f max %.2f; lockstep %d mean %.2f min %.2f max %.2f\n"
                         % (k, len(v), sum(d for _, d in v) / len(v) / 1e3, len(o), sum(o) / max(len(o), 1) / 1e3, min(o) / 1e3, max(o) / 1e3,
                            len(l), sum(l) / max(len(l), 1) / 1e3, (min(l) if l else 0) / 1e3, (max(l) if l else 0) / 1e3))


if __name__ == "__main__":
    main()
