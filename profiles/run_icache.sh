#!/bin/bash
# Instruction-cache counters of the kernels of a ./ba run on a small sequence (GPU box): is k_persist, whose loop body is
# larger than the 64 KB instruction cache two CUs share, fetch bound?    bash profiles/run_icache.sh [fr1xyz] [extra ba flags]
SEQ=${1:-fr1xyz}
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/icache_$SEQ
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o run --output-format csv -- $REPO/gbp_poplar_amd/bin/ba --bal_file $REPO/data/sequences/$SEQ.txt --eval_every 100 "$@" > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print("   %-30s %.4g" % (c, tot[k][c]))
PY
