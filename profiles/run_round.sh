#!/bin/bash
# Round summary runs (GPU box, repo root):  bash profiles/run_round.sh r02   -> gpurun_out/round_<tag>/
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round_$TAG
mkdir -p $OUT
cd $REPO
bash profiles/run_profile.sh $TAG > $OUT/run_profile.log 2>&1
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pmc off > $OUT/bench_s1_200.json 2> /dev/null
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pmc off --cams 8000 --lmks 125000 > $OUT/bench_c5shape_plain.json 2> /dev/null
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --force-sharded --cams 8000 --lmks 125000 > $OUT/bench_c5shape_native.json 2> /dev/null
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --force-sharded --comm torch --sharded-graph 0 --cams 8000 --lmks 125000 > $OUT/bench_c5shape_torch.json 2> /dev/null
# the line an N > 1 run prints, on one rank: PMC passes on the shard shape, exchange timing
python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_c5shape_driverline.json 2> /dev/null
for seq in fr1xyz fr2robot2 fr1desk; do
  for ev in 1 100; do
    for rep in 1 2; do gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt --eval_every $ev > $OUT/ba_${seq}_every$ev.log 2>&1; done
  done
done
gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt > $OUT/slam_fr2robot2.log 2>&1
gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt --eval_every 100 > $OUT/slam_fr2robot2_every100.log 2>&1
for seq in fr1xyz fr2robot2; do GBP_PERSIST=-1 gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt --eval_every 100 > $OUT/ba_${seq}_every100_twokernels.log 2>&1; done
gbp_poplar_amd/bin/ba --bal_file data/sequences/fr2robot2.txt --ipus 2 > $OUT/ba_fr2robot2_ipus2.log 2>&1
python3 - <<PY
import json, glob, os, re
out = "$OUT"
print("| run | value (1M-factor it/s) | ms/step | sweep us | beliefs us | graph | exchange |")
print("|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(out + "/bench_*.json")) + [out + "/../prof_$TAG/bench.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print("|", os.path.basename(f), "| failed", e, "|"); continue
    r = d["roofline"]
    print("| %s | %.1f | %.4f | %s | %s | %s | %s | frac %s exch %s us |" % (os.path.basename(f), d["value"], d["ms_per_step"], r["avg_launch_us"], r["belief_kernels_avg_us"],
          d["config"].get("iteration_graph"), (d["config"].get("exchange") or "-")[:40], r.get("frac"), r.get("exchange_avg_us")))
print()
for f in sorted(glob.glob(out + "/*.log")):
    t = open(f).read()
    m = re.findall(r"Total time: .*", t)
    last = [l for l in t.splitlines() if l.startswith(("Iter ", "Iters "))]
    if m:
        print("%-28s %s | %s" % (os.path.basename(f), m[-1][:150], last[-1][:110] if last else ""))
PY
