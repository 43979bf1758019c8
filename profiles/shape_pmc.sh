#!/bin/bash
# HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes: the guide's recipe) and the time per iteration
# of variants of the device order / cache policy on one graph shape:
#   SHAPE="8000 125000" bash profiles/shape_pmc.sh <tag> "classes=16" "classes=16 row_key_lane=8" "policy=6" ...
#   -> gpurun_out/<tag>_shape.txt     (one line per variant: us per iteration, k_sweep MB, k_beliefs MB; mean over the last 40 dispatches)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_shape
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT.txt
for v in "" "$@"; do
  name=$(echo "${v:-product}" | tr ' =' '_-')
  t=$(python3 $R/profiles/shape_variant.py ${SHAPE:-8000 125000} $v 2>/dev/null | tail -1)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/$name-$c
    rocprofv3 --pmc $c --output-format csv -d $OUT/$name-$c -o p -- python3 $R/profiles/shape_variant.py ${SHAPE:-8000 125000} $v direct=1 iters=40 > /dev/null 2>&1
  done
  python3 - "$OUT/$name" "$t" >> $OUT.txt <<'PY'
import csv, glob, sys
base, t = sys.argv[1], sys.argv[2]
def mean(counter, kern):
    v = []
    for f in glob.glob(base + "-" + counter + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]:
                v.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    v.sort()
    v = [x for _, x in v[-40:]]
    return sum(v) / len(v) if v else float("nan")
out = []
for k in ("k_sweep", "k_beliefs"):
    f, w = mean("FETCH_SIZE", k), mean("WRITE_SIZE", k)
    out.append("%s %.1f MB (fetch x2 %.1f + write %.1f)" % (k, (2 * f + w) / 1024, 2 * f / 1024, w / 1024))
print(t, "|", " | ".join(out))
PY
  rm -rf $OUT/$name-FETCH_SIZE $OUT/$name-WRITE_SIZE
done
cat $OUT.txt
