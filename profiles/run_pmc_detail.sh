#!/bin/bash
# Extra PMC passes for the two iteration kernels on S1 (one counter group per pass, kernel-trace off):
#   bash profiles/run_pmc_detail.sh r01
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 12 --cpu-seconds 0 --profile-steps 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -o tcc -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/tcc.log
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/ea -o ea -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/ea.log
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RD_UNCACHED_32B_sum --output-format csv -d $OUT/ea2 -o ea2 -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/ea2.log
python3 - <<PY
import csv, glob, os
out = "$OUT"
lines = ["# PMC detail ($TAG): mean per dispatch", ""]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    agg = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_sweep" not in k and "k_beliefs" not in k:
            continue
        key = ("k_sweep" if "k_sweep" in k else "k_beliefs", r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    lines += ["## " + os.path.relpath(f, out), "", "| kernel | counter | dispatches | mean |", "|---|---|---|---|"]
    for (k, c), (n, s) in sorted(agg.items()):
        lines.append("| %s | %s | %d | %.4g |" % (k, c, n, s / n))
    lines.append("")
open(os.path.join("$REPO", "gpurun_out", "pmc_detail_$TAG.md"), "w").write("\n".join(lines))
print("\n".join(lines))
PY
tail -3 $OUT/sq.log $OUT/tcc.log $OUT/ea.log
