#!/bin/bash
# Round 4, second GPU pass: GPU suite, bench line, cooperative vs plain k_persist launch, persist trace, per-dispatch PMC of S1.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04b
mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "sweep", d["roofline"]["avg_launch_us"], "beliefs", d["roofline"]["belief_kernels_avg_us"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("configs"), indent=1)[:6000])
PY
BA=gbp_poplar_amd/bin/ba; SLAM=gbp_poplar_amd/bin/slam
for coop in 0 -1; do
  echo "== GBP_PERSIST_COOP=$coop"
  for ev in 100 1; do for rep in 1 2 3; do
    GBP_PERSIST_COOP=$coop $BA --bal_file data/sequences/fr1xyz.txt --eval_every $ev 2>&1 | grep -E "Total time|warning" | cut -c1-230
  done; done
  GBP_PERSIST_COOP=$coop $BA --bal_file data/sequences/fr2robot2.txt --eval_every 100 2>&1 | grep -E "Total time|warning" | cut -c1-230
  GBP_PERSIST_COOP=$coop $SLAM --bal_file data/sequences/fr2robot2.txt --eval_every 100 2>&1 | grep -E "Total time|warning" | cut -c1-230
  GBP_PERSIST_COOP=$coop $SLAM --bal_file data/sequences/fr2robot2.txt 2>&1 | grep -E "Total time|warning" | cut -c1-230
done
echo "== two processes at once, plain launches"
for i in 1 2 3; do
  (GBP_PERSIST_COOP=-1 $BA --bal_file data/sequences/fr1xyz.txt --eval_every 100 2>&1 | grep -E "Total time|warning|Iter 1499" | cut -c1-200) &
  (GBP_PERSIST_COOP=-1 $BA --bal_file data/sequences/fr1xyz.txt --eval_every 100 2>&1 | grep -E "Total time|warning|Iter 1499" | cut -c1-200) &
  wait
done
echo "== two processes at once, cooperative launches"
for i in 1 2 3; do
  ($BA --bal_file data/sequences/fr1xyz.txt --eval_every 100 2>&1 | grep -E "Total time|warning|Iter 1499" | cut -c1-200) &
  ($BA --bal_file data/sequences/fr1xyz.txt --eval_every 100 2>&1 | grep -E "Total time|warning|Iter 1499" | cut -c1-200) &
  wait
done
echo "== persist trace"
timeout 300 python3 profiles/persist_trace.py fr1xyz > $OUT/persist_trace_fr1xyz.txt 2>&1; tail -40 $OUT/persist_trace_fr1xyz.txt
timeout 300 python3 profiles/persist_trace.py fr1xyz each > $OUT/persist_trace_fr1xyz_each.txt 2>&1; tail -25 $OUT/persist_trace_fr1xyz_each.txt
echo "== per-dispatch trace + PMC of S1 (lock-step sweep 17 inside)"
cd /tmp && export TMPDIR=/tmp
CHILD="$REPO/bench.py --pmc-child --steps 14 --warmup 12"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ld_trace -o t -- python3 $CHILD > /dev/null 2> $OUT/ld_trace.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ld_fetch -o f -- python3 $CHILD > /dev/null 2> $OUT/ld_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ld_write -o w -- python3 $CHILD > /dev/null 2> $OUT/ld_write.log
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $OUT/ld_sq -o s -- python3 $CHILD > /dev/null 2> $OUT/ld_sq.log
cd $REPO
python3 profiles/relin_dispatches.py $OUT/ld_trace $OUT/ld_fetch $OUT/ld_write > $OUT/relin_dispatches.csv 2> $OUT/relin_dispatches.txt
cat $OUT/relin_dispatches.txt; grep k_sweep $OUT/relin_dispatches.csv | head -30
python3 - <<PY
import csv, glob
agg = {}
for f in glob.glob("$OUT/ld_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep" in r["Kernel_Name"]:
            agg.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (d, v) in enumerate(sorted(agg.items())):
    print(i, {k: "%.3g" % x for k, x in sorted(v.items())})
PY
