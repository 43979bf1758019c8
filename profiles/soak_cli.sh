#!/bin/bash
# Repeatability of the CLI flows through k_persist (GPU box): N runs each, the metric lines of every run must hash alike — and
# like the two-kernel path's (GBP_PERSIST=-1: the reference line of each configuration).
#   bash profiles/soak_cli.sh [N=60]
N=${1:-60}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for seq in fr1xyz fr2robot2 fr1desk; do
  for mode in "" "--eval_every 100"; do
    echo -n "ba $seq $mode two-kernel path: "; GBP_PERSIST=-1 gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt $mode 2>&1 | grep "^Iter\|Weakening\|warning" | md5sum
    for i in $(seq $N); do gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt $mode 2>&1 | grep "^Iter\|Weakening\|warning" | md5sum; done | sort | uniq -c | sed "s|^|ba $seq $mode: |"
  done
done
for mode in "" "--eval_every 100"; do
  echo -n "slam fr2robot2 $mode two-kernel path: "; GBP_PERSIST=-1 gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt $mode 2>&1 | grep -v "Total time" | md5sum
  for i in $(seq $((N / 4))); do gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt $mode 2>&1 | grep -v "Total time" | md5sum; done | sort | uniq -c | sed "s|^|slam fr2robot2 $mode: |"
done
