#!/bin/bash
# Executed-instruction mix of k_persist per wave and iteration on a shipped sequence (bursts of 100 iterations, no metric):
#   [LD_LIBRARY_PATH=<variant dir>] bash profiles/inst_mix_persist.sh <tag> [fr1xyz]  -> gpurun_out/<tag>_inst_mix_persist.txt
TAG=${1:-r04}; SEQ=${2:-fr1xyz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_imp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o c -- $R/gbp_poplar_amd/bin/ba --bal_file $R/data/sequences/$SEQ.txt --eval_every 100 > /dev/null 2> $OUT/p$i.err
done
python3 - $OUT > $R/gpurun_out/${TAG}_inst_mix_persist.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
per = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_persist" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        per.setdefault(r["Counter_Name"], {})[ids.index(int(r["Dispatch_Id"]))] = float(r["Counter_Value"])
n = len(next(iter(per.values())))
d = n - 2                                   # a late burst of 100 iterations
waves = per["SQ_WAVES"][d]
print("k_persist dispatch %d of %d: %.0f waves; per wave and iteration (100 iterations per launch)" % (d, n, waves))
for k, v in per.items():
    print("%s | %.1f" % (k, v[d] / waves / 100.0))
PY
rm -rf $OUT
cat $R/gpurun_out/${TAG}_inst_mix_persist.txt
