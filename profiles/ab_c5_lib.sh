#!/bin/bash
# A/B of product-library variants on the config-5 shard shape (plain ctx and the sharded line), interleaved on one box:  bash profiles/ab_c5_lib.sh w7 w8
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in 1 2 3; do for v in "$@"; do for mode in "" "--force-sharded"; do
  GBP_LIB=$REPO/profiles/_bin/$v/libgbp_mi355x.so python3 bench.py --gpus 1 $mode --cams 8000 --lmks 125000 --steps 200 --warmup 20 --cpu-seconds 0 --small-configs off --pmc off 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '${mode:-plain}', 'value %.1f | windows %.1f / %.1f / %.1f | sustained %.1f | sweep %.2f us | beliefs %s us' % (d['value'], d['windows']['min'], d['windows']['median'], d['windows']['max'], d['sustained']['value'], r['avg_launch_us'], r['belief_kernels_avg_us']))"
done; done; done
