#!/bin/bash
TAG=${1:-r04i}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "sweep", d["roofline"]["avg_launch_us"], "beliefs", d["roofline"]["belief_kernels_avg_us"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for k, c in (d.get("configs") or {}).items():
    print(k, {x: c.get(x) for x in ("iters_per_sec", "loop_wall_ms", "device_ms", "us_per_iter_device", "final_mean_reproj_px", "graph_state", "error")}, c.get("eval_every_100"), (c.get("cpu_baseline") or {}).get("value"))
PY
timeout 600 python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_c5shape_driverline.json 2> $OUT/bench_c5.err; echo "bench c5 rc $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_c5shape_driverline.json").read().strip().splitlines()[-1])
print("c5 shard shape value", d["value"], "ms/step", d["ms_per_step"], "exchange", d["roofline"]["exchange_avg_us"])
print(json.dumps(d["config"]["preflight"], indent=1))
PY
