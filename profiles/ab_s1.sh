#!/bin/bash
# A/B of product-library variants on the S1 bench line (one box, alternating rounds; rule: never compare across devices):
#   VARIANTS="default dppfused" bash profiles/ab_s1.sh <tag> [steps] [warmup]  -> gpurun_out/<tag>/run.log
TAG=${1:-r04o}; STEPS=${2:-200}; WARM=${3:-20}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
python3 bench.py --steps 50 --warmup 10 --pmc off --cpu-seconds 0 --small-configs off > /dev/null 2>&1   # warm the box
for round in $(seq 1 ${ROUNDS:-4}); do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then L=""; else L=$REPO/profiles/_bin/$v/libgbp_mi355x.so; fi
  echo -n "$v round $round: "
  GBP_LIB=$L python3 bench.py --steps $STEPS --warmup $WARM --pmc off --cpu-seconds 0 --small-configs off 2> $OUT/err.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={x['kernel'].split('<')[0].split('(')[0]:x for x in d['roofline'].get('kernels',[])} if isinstance(d['roofline'].get('kernels'),list) else {}
print(d['value'], d['ms_per_step'], 'sweep', d['roofline'].get('avg_launch_us'), 'beliefs', d['roofline'].get('belief_kernels_avg_us'))
"
done
done
