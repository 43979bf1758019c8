#!/bin/bash
TAG=${1:-r04m}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_experiments.py tests/test_cli.py -m gpu -q -x -k "persistent or iterate_eval or health or config1 or config3 or cli or two_processes or ipus or time_out or degenerate" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
VARIANTS="default" bash profiles/run_r04d.sh $TAG
cat $OUT/run.log
timeout 300 python3 profiles/persist_trace.py fr1xyz each > $OUT/persist_trace_fr1xyz_each.txt 2>&1; head -24 $OUT/persist_trace_fr1xyz_each.txt
# drift of k_sweep over a long run, both tile orders
cd /tmp && export TMPDIR=/tmp
for to in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/drift_to$to -o t -- python3 $REPO/bench.py --pmc-child --steps 150 --warmup 12 --tile-order $to > /dev/null 2> $OUT/drift_to$to.log
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/drift_to$to/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k_sweep" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("tile_order $to k_sweep us by dispatch (groups of 10, lock-step > 130 excluded):", [round(sum(x for x in d[i:i+10] if x < 130) / max(1, sum(1 for x in d[i:i+10] if x < 130)), 1) for i in range(0, len(d), 10)])
PY
done
