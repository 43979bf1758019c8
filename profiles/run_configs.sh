#!/bin/bash
# BASELINE.json configs 1-3 on one MI355X through the reference-compatible CLIs (run from the repo root on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/configs
mkdir -p $OUT
BA=$REPO/gbp_poplar_amd/bin/ba; SLAM=$REPO/gbp_poplar_amd/bin/slam
for seq in fr1xyz fr2robot2 fr1desk; do
  $BA --bal_file $REPO/data/sequences/$seq.txt --n_iters 1500 > $OUT/ba_${seq}_every1.log 2>&1
  $BA --bal_file $REPO/data/sequences/$seq.txt --n_iters 1500 --eval_every 100 > $OUT/ba_${seq}_every100.log 2>&1
done
$SLAM --bal_file $REPO/data/sequences/fr2robot2.txt > $OUT/slam_fr2robot2.log 2>&1
for f in $OUT/*.log; do echo "== $f"; grep -E "Initial Reprojection|Total time" $f; grep -E "^Iter" $f | tail -2; done
