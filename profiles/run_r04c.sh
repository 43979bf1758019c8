#!/bin/bash
# Round 4, third GPU pass.  Everything goes to files under gpurun_out/<tag>/ (the caller's terminal only sees a tail).
TAG=${1:-r04c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "sweep", d["roofline"]["avg_launch_us"], "beliefs", d["roofline"]["belief_kernels_avg_us"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for k, c in (d.get("configs") or {}).items():
    print(k, {x: c.get(x) for x in ("iters_per_sec", "loop_wall_ms", "device_ms", "us_per_iter_device", "final_mean_reproj_px", "graph_state", "error")}, c.get("eval_every_100"), (c.get("cpu_baseline") or {}).get("value"))
PY
BA=gbp_poplar_amd/bin/ba; SLAM=gbp_poplar_amd/bin/slam
for coop in 0 1; do
  echo "== GBP_PERSIST_COOP=$coop"
  for ev in 100 1; do for rep in 1 2 3; do
    GBP_PERSIST_COOP=$coop $BA --bal_file data/sequences/fr1xyz.txt --eval_every $ev 2>&1 | grep -E "Total time|warning" | cut -c1-230
  done; done
  GBP_PERSIST_COOP=$coop $BA --bal_file data/sequences/fr2robot2.txt --eval_every 100 2>&1 | grep -E "Total time|warning" | cut -c1-230
  GBP_PERSIST_COOP=$coop $BA --bal_file data/sequences/fr1desk.txt --eval_every 100 2>&1 | grep -E "Total time|warning" | cut -c1-230
  GBP_PERSIST_COOP=$coop $SLAM --bal_file data/sequences/fr2robot2.txt --eval_every 100 2>&1 | grep -E "Total time|warning" | cut -c1-230
  GBP_PERSIST_COOP=$coop $SLAM --bal_file data/sequences/fr2robot2.txt 2>&1 | grep -E "Total time|warning" | cut -c1-230
done
echo "== persist trace"
timeout 300 python3 profiles/persist_trace.py fr1xyz > $OUT/persist_trace_fr1xyz.txt 2>&1; head -24 $OUT/persist_trace_fr1xyz.txt
timeout 300 python3 profiles/persist_trace.py fr1xyz each > $OUT/persist_trace_fr1xyz_each.txt 2>&1; head -24 $OUT/persist_trace_fr1xyz_each.txt
echo "== per-dispatch trace + PMC of S1 (lock-step sweep inside)"
cd /tmp && export TMPDIR=/tmp
CHILD="$REPO/bench.py --pmc-child --steps 24 --warmup 12"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ld_trace -o t -- python3 $CHILD > /dev/null 2> $OUT/ld_trace.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ld_fetch -o f -- python3 $CHILD > /dev/null 2> $OUT/ld_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ld_write -o w -- python3 $CHILD > /dev/null 2> $OUT/ld_write.log
cd $REPO
python3 profiles/relin_dispatches.py $OUT/ld_trace $OUT/ld_fetch $OUT/ld_write > $OUT/relin_dispatches.csv 2> $OUT/relin_dispatches.txt
cat $OUT/relin_dispatches.txt; grep -E "lockstep" $OUT/relin_dispatches.csv
cp $OUT/ld_trace/t_kernel_stats.csv $OUT/kernel_stats_ld.csv 2>/dev/null
