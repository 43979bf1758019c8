#!/bin/bash
# Counters per k_sweep dispatch of relin_hump_b2b.py (one pass of 80 sweeps is enough: 3 x 80 dispatches are recorded):
#   bash profiles/relin_hump_pmc.sh   -> gpurun_out/r04_hump_pmc/<set>.csv  (dispatch order = sweep order)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_hump_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o c -- python3 $R/profiles/relin_hump_b2b.py 80 > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$OUT/set$i.csv" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sweep" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in by.values() for k in v})
with open(sys.argv[2], "w") as o:
    o.write("k_sweep_dispatch," + ",".join(names) + "\n")
    for i, (d, v) in enumerate(sorted(by.items())):
        o.write("%d,%s\n" % (i, ",".join("%.0f" % v.get(n, -1) for n in names)))
PY
  rm -rf $OUT/p$i
done
ls -la $OUT
