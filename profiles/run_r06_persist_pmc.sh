#!/bin/bash
# Round 6 item 4: what does a wave of the persistent kernel spend its time on?  SQ counters of the default `ba fr1xyz` / `slam fr2robot2` runs
#   WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES  (MI355X_MICROARCH.md)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$REPO/gpurun_out/r06_persist_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/ba_$n -o c -- $REPO/gbp_poplar_amd/bin/ba --bal_file $REPO/data/sequences/fr1xyz.txt > $OUT/ba_$n.log 2>&1
  rocprofv3 --pmc $set --output-format csv -d $OUT/ba100_$n -o c -- $REPO/gbp_poplar_amd/bin/ba --bal_file $REPO/data/sequences/fr1xyz.txt --eval_every 100 > $OUT/ba100_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for pre in ("ba_", "ba100_"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/" + pre + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gbp::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "persist" not in k: continue
        wc = v.get("SQ_WAVE_CYCLES", 0)
        print(pre, k, {n: int(x) for n, x in sorted(v.items())})
        if wc:
            print("    of the wave cycles: parked (WAIT_ANY) %.1f %%, issue stall (WAIT_INST_ANY) %.1f %%, issuing (ACTIVE_INST_ANY) %.1f %%" % (100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc))
PY
