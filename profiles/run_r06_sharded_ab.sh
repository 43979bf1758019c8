#!/bin/bash
# A/B of the sharded line on the config-5 shard shape (1-rank communicator): direct launches vs the captured graph, 3 runs each, beside the plain ctx
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_sharded_ab
mkdir -p $OUT
cd $REPO
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off 2> /dev/null | grep '^{' > $OUT/direct_$rep.json
  python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --sharded-graph 1 2> /dev/null | grep '^{' > $OUT/graph_$rep.json
  python3 bench.py --gpus 1 --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --small-configs off 2> /dev/null | grep '^{' > $OUT/plain_$rep.json
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    pre = d["config"].get("preflight") or {}
    print(f.split("/")[-1], "first %.4f ms | windows med %.4f ms | sustained %.4f ms | sched %s | graph %s" % (
        d["ms_per_step"], d["config"]["factors"] / 1e6 / d["windows"]["median"] * 1e3, d["config"]["factors"] / 1e6 / d["sustained"]["value"] * 1e3,
        pre.get("schedule_ms_per_iteration"), d["config"].get("iteration_graph")))
PY
