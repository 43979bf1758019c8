#!/bin/bash
# Round 6 item 5: the config-5 shard shape (8 000 x 125 000 x 1.25 M) with and without the all-pad segments skipped: PMC traffic + time
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_seg
mkdir -p $OUT
cd $REPO
for rep in 1 2; do for mode in 0 1; do
  GBP_SEG_SKIP=$mode python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --small-configs off --cams 8000 --lmks 125000 2> $OUT/seg${mode}_$rep.err | grep '^{' > $OUT/seg${mode}_$rep.json
done; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/seg*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; rep = r.get("replay") or {}
    print(f.split("/")[-1], "value %.1f  ms/step %.4f | windows med %.1f | sustained %.1f | sweep %.2f us | traffic %s (ordinary %s, lock-step %s) | t/layout %s | frac %s" % (
        d["value"], d["ms_per_step"], d["windows"]["median"], d["sustained"]["value"], r["avg_launch_us"], r["traffic"], rep.get("traffic_ordinary_launch"), rep.get("traffic_lockstep_launch"), r["traffic_over_layout"], r["frac"]))
PY
