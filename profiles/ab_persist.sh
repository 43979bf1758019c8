#!/bin/bash
# A/B of the persistent kernel on ONE box: variant libraries under profiles/_bin/<name>/ (profiles/build_variant.sh), interleaved
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
VARS=${@:-base}
for rep in 1 2 3; do
  for seq in fr1xyz fr2robot2; do
    for v in $VARS; do
      echo "$v $(GBP_LIB=$REPO/profiles/_bin/$v/libgbp_mi355x.so python3 profiles/time_bursts.py $seq 20 2>/dev/null | tail -1 | sed 's/ on the device//g; s/(graph_state 2) //')"
    done
  done
done
