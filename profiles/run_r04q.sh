#!/bin/bash
TAG=${1:-r04q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
for rep in 1 2; do
for gu in 10 20 40; do
  for st in 20 200; do
  python3 bench.py --graph-unroll $gu --steps $st --warmup 20 --cpu-seconds 0 --small-configs off --pmc off --profile-steps 0 > $OUT/b.json 2> /dev/null
  python3 -c "
import json
d = json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('graph_unroll $gu steps $st: value', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
done
