#!/bin/bash
# Executed-instruction mix of k_sweep per wave (ordinary sweep 5 and the lock-step sweep 18 of the ./ba flow on S1):
#   [GBP_LIB=<variant .so>] bash profiles/inst_mix.sh <tag> [k_sweep|k_beliefs]  -> gpurun_out/<tag>_inst_mix.txt
TAG=${1:-r04}; export KERN=${2:-k_sweep}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_inst_mix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VSKIPPED SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o c -- python3 $R/profiles/relin_hump_b2b.py 20 > /dev/null 2> $OUT/p$i.err
done
python3 - $OUT > $R/gpurun_out/${TAG}_inst_mix.txt <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
per = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if os.environ.get("KERN", "k_sweep") in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        per.setdefault(r["Counter_Name"], {})[ids.index(int(r["Dispatch_Id"]))] = float(r["Counter_Value"])
waves = per.get("SQ_WAVES", {}).get(5, 15625.0)
print("counter | ordinary sweep (dispatch 5), per wave | lock-step sweep (dispatch 18), per wave")
for k, v in per.items():
    print("%s | %.1f | %.1f" % (k, v.get(5, -1) / waves, v.get(18, -1) / waves))
PY
rm -rf $OUT
cat $R/gpurun_out/${TAG}_inst_mix.txt
