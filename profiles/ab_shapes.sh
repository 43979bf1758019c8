#!/bin/bash
# A/B of product-library variants over graph SHAPES (one box, alternating; the cache-policy scans of profiles/r04_alu_diet.md section 6):
#   VARIANTS="oldpol default" SHAPES="1000x100000 8000x125000 1000x200000" ROUNDS=3 bash profiles/ab_shapes.sh <tag>
#   -> gpurun_out/<tag>_shapes.txt   (shape = cameras x landmarks, 10 factors per landmark; variants: profiles/build_variant.sh)
TAG=${1:-shapes}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/${TAG}_shapes.txt
: > $OUT
for sh in ${SHAPES:-1000x100000}; do
  c=${sh%x*}; l=${sh#*x}
  for r in $(seq 1 ${ROUNDS:-3}); do for v in ${VARIANTS:-default}; do
    if [ $v = default ]; then L=""; else L=$PWD/profiles/_bin/$v/libgbp_mi355x.so; fi
    echo -n "$v $sh: " >> $OUT
    GBP_LIB=$L python3 bench.py --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --pmc off --small-configs off --cams $c --lmks $l 2>/dev/null | tail -1 \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT
  done; done
done
awk '{k=$1" "$2; n[k]++; s[k]+=$3} END{for (v in n) printf "AVG %s %.0f\n", v, s[v]/n[v]}' $OUT | sort -k3
