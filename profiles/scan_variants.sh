#!/bin/bash
# Executed VALU instructions per k_sweep wave (ordinary sweep 5 / lock-step sweep 18) for a list of product-library variants:
#   VARIANTS="default fbp ..." bash profiles/scan_variants.sh <tag>  -> gpurun_out/<tag>_scan.txt
TAG=${1:-scan}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_scan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then L=""; else L=$R/profiles/_bin/$v/libgbp_mi355x.so; fi
  rm -rf $OUT/p
  GBP_LIB=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/p -o c -- python3 $R/profiles/relin_hump_b2b.py 20 > /dev/null 2> $OUT/$v.err
  python3 - $OUT/p $v >> $R/gpurun_out/${TAG}_scan.txt <<'PY'
import csv, glob, sys, collections
per = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_sweep" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        per.setdefault(r["Counter_Name"], {})[ids.index(int(r["Dispatch_Id"]))] = float(r["Counter_Value"])
w = per["SQ_WAVES"][5]
print("%s: VALU per wave %.0f ordinary, %.0f lock-step; SALU %.0f" % (sys.argv[2], per["SQ_INSTS_VALU"][5] / w, per["SQ_INSTS_VALU"][18] / w, per["SQ_INSTS_SALU"][5] / w))
PY
done
rm -rf $OUT
cat $R/gpurun_out/${TAG}_scan.txt
