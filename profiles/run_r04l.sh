#!/bin/bash
TAG=${1:-r04l}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], ": value", d["value"], "ms/step", d["ms_per_step"], "sweep", r["avg_launch_us"], "beliefs", r["belief_kernels_avg_us"], "exch", r.get("exchange_avg_us"), "traffic MB", round(r["traffic"]/1e6,1) if r.get("traffic") else None, (d["config"].get("preflight") or {}).get("schedule_ms_per_iteration"))
PY
}
for to in 0 1; do
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pmc off --cams 8000 --lmks 125000 --tile-order $to --small-configs off > $OUT/c5_plain_to$to.json 2> /dev/null; show $OUT/c5_plain_to$to.json "c5 shape plain tile_order $to"
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pmc off --force-sharded --cams 8000 --lmks 125000 --tile-order $to > $OUT/c5_sharded_to$to.json 2> /dev/null; show $OUT/c5_sharded_to$to.json "c5 shape sharded tile_order $to"
done
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pmc off --force-sharded --cams 8000 --lmks 125000 --preflight 0 > $OUT/c5_sharded_nopre.json 2> /dev/null; show $OUT/c5_sharded_nopre.json "c5 shape sharded no preflight"
