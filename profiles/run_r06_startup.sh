#!/bin/bash
# Round 6 item 6: where does the wall time of `bin/ba fr1xyz` go?  5 runs each of ba fr1xyz / slam fr2robot2 with --profile 1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$REPO/gpurun_out/r06_startup_$TAG
mkdir -p $OUT
cd $REPO
for rep in 1 2 3 4 5; do
  for cfg in "ba fr1xyz" "slam fr2robot2" "ba fr2robot2"; do
    set -- $cfg
    mkdir -p $OUT/p_$1_$2_$rep
    s=$(date +%s.%N)
    GC_PROFILE_LOG_DIR=$OUT/p_$1_$2_$rep gbp_poplar_amd/bin/$1 --bal_file data/sequences/$2.txt --profile 1 > $OUT/$1_$2_$rep.log 2>&1
    e=$(date +%s.%N)
    echo "$1 $2 run $rep: process wall $(echo "$e - $s" | bc) s :: $(grep 'Total time' $OUT/$1_$2_$rep.log | cut -c1-330)"
  done
done | tee $OUT/summary.txt
for f in $OUT/ba_fr1xyz_*.log $OUT/slam_fr2robot2_*.log; do echo "$(basename $f) $(grep -v 'Total time\|Profile written' $f | md5sum | cut -c1-32)"; done > $OUT/cli_md5.txt
cat $OUT/p_ba_fr1xyz_5/gbp_profile.json
sort -k2 $OUT/cli_md5.txt | uniq -c -f1 | head
cat profiles/r05_cli_md5.txt
