#!/bin/bash
# A/B of product-library variants on the small-graph bursts (time_bursts.py: k_persist us/iteration), one box, alternating:
#   VARIANTS="rowplain default" bash profiles/ab_small.sh <tag>  -> gpurun_out/<tag>/small.log
TAG=${1:-r04w}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/small.log 2>&1
python3 profiles/time_bursts.py fr1xyz 3 > /dev/null 2>&1
for round in $(seq 1 ${ROUNDS:-3}); do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then L=""; else L=$REPO/profiles/_bin/$v/libgbp_mi355x.so; fi
  for seq in fr1xyz fr2robot2; do
    echo -n "$v round $round: "; GBP_LIB=$L python3 profiles/time_bursts.py $seq 10 2>/dev/null | tail -1
  done
done
done
