cd /root/repo
for seq in fr1xyz fr2robot2 fr1desk; do for rep in 1 2 3; do gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt > /tmp/ba_$seq.log 2>&1; done; grep "Total time" /tmp/ba_$seq.log | cut -c1-200; grep "^Iter\|Weakening" /tmp/ba_$seq.log | md5sum; done
for rep in 1 2; do gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt > /tmp/slam.log 2>&1; done; grep "Total time" /tmp/slam.log | cut -c1-200; grep -v "Total time" /tmp/slam.log | md5sum
