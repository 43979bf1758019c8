#!/bin/bash
# A/B of product-library variants on the default loop (profiles/time_default_loop.py), one box, alternating rounds:
#   VARIANTS="default evx_a evx_b" [SHAPE="1000 100000"] bash profiles/ab_default_loop.sh <tag>  -> gpurun_out/<tag>_loop.txt
TAG=${1:-loop}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/${TAG}_loop.txt
: > $OUT
for r in $(seq 1 ${ROUNDS:-2}); do for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then L=""; else L=$PWD/profiles/_bin/$v/libgbp_mi355x.so; fi
  GBP_LIB=$L python3 profiles/time_default_loop.py ${SHAPE:-1000 100000} 200 2>/dev/null | grep "gbp_iterate" | sed "s/^/$v /" >> $OUT
done; done
cat $OUT
