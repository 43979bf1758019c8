#!/usr/bin/env python3
"""k_sweep durations per pass and sweep from the kernel trace of relin_hump_b2b.py:  python3 profiles/relin_hump_b2b_report.py <dir> [sweeps]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_sweep" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t0 = [int(r["Start_Timestamp"]) for r in rows]
assert len(dur) == 3 * n, (len(dur), n)
print("sweep,pass1_us,pass2_us,pass3_us,pass1_ms_since_its_first_sweep")
for i in range(n):
    print("%d,%.2f,%.2f,%.2f,%.3f" % (i, dur[i], dur[n + i], dur[2 * n + i], (t0[i] - t0[0]) / 1e6))
