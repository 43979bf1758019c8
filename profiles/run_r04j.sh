#!/bin/bash
TAG=${1:-r04j}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
for rep in 1 2; do
for to in 0 1 3; do
  timeout 600 python3 bench.py --tile-order $to --steps 200 --warmup 20 --cpu-seconds 0 --small-configs off > $OUT/bench_to$to.json 2> $OUT/bench_to$to.err
  python3 - <<PY
import json
d = json.loads(open("$OUT/bench_to$to.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("tile_order $to: value", d["value"], "ms/step", d["ms_per_step"], "sweep", r["avg_launch_us"], "beliefs", r["belief_kernels_avg_us"], "traffic MB", round(r["traffic"]/1e6,1) if r["traffic"] else None, "frac", r["frac"], "rmse", d["config"]["reproj_rmse_px_final"])
PY
done
done
