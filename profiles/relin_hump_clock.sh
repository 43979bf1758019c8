#!/bin/bash
# Shader clock per k_sweep dispatch of relin_hump_b2b.py: GRBM_GUI_ACTIVE cycles / (8 XCDs x duration), and the counter list.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_hump_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/q1 -o c -- python3 $R/profiles/relin_hump_b2b.py 80 > /dev/null 2> $OUT/q1.err
f=$(find $OUT/q1 -name '*counter_collection.csv' | head -1)
python3 - "$f" > $OUT/clock.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sweep" in r["Kernel_Name"]]
print("k_sweep_dispatch,duration_us,GRBM_GUI_ACTIVE,GHz_if_summed_over_8_XCDs")
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = float(r["Counter_Value"])
    print("%d,%.2f,%.0f,%.3f" % (i, d, c, c / 8 / d / 1e3))
PY
rm -rf $OUT/q1
