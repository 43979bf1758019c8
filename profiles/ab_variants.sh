#!/bin/bash
# A/B of product-library variants through the CLIs (same box, same run, alternating): profiles/_bin/<variant>/libgbp_mi355x.so built by
# profiles/build_variant.sh; VARIANTS="default rows3 ..." bash profiles/ab_variants.sh <tag>  -> gpurun_out/<tag>/run.log
TAG=${1:-r04d}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
BA=gbp_poplar_amd/bin/ba; SLAM=gbp_poplar_amd/bin/slam
$BA --bal_file data/sequences/fr1xyz.txt --eval_every 100 > /dev/null 2>&1   # warm the box
for round in 1 2 3; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then L=""; else L=$REPO/profiles/_bin/$v; fi
  for ev in 100 1; do
    echo -n "$v fr1xyz every$ev: "; LD_LIBRARY_PATH=$L $BA --bal_file data/sequences/fr1xyz.txt --eval_every $ev 2>&1 | grep -E "Total time" | sed 's/.*device time in GBP iterations: //' | cut -c1-60
  done
  echo -n "$v fr2robot2 every100: "; LD_LIBRARY_PATH=$L $BA --bal_file data/sequences/fr2robot2.txt --eval_every 100 2>&1 | grep -E "Total time" | sed 's/.*device time in GBP iterations: //' | cut -c1-60
  echo -n "$v slam default: "; LD_LIBRARY_PATH=$L $SLAM --bal_file data/sequences/fr2robot2.txt 2>&1 | grep -E "Total time" | sed 's/.*device time in GBP iterations: //' | cut -c1-60
done
done
