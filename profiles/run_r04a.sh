#!/bin/bash
# Round 4, first GPU pass: the whole GPU suite, the driver's bench line, per-dispatch kernel trace of S1, CLI timings.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04a
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
tail -c 3000 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py --steps 60 --warmup 20 --cpu-seconds 0 --pmc off --small-configs off > $OUT/trace_bench.json 2> $OUT/trace.log
cd $REPO
python3 profiles/relin_dispatches.py $OUT/trace > $OUT/relin_dispatches.csv 2> $OUT/relin_dispatches.txt
cat $OUT/relin_dispatches.txt
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
for ev in 100 1; do for rep in 1 2 3; do
  gbp_poplar_amd/bin/ba --bal_file data/sequences/fr1xyz.txt --eval_every $ev > $OUT/ba_fr1xyz_every$ev.log 2>&1
  grep "Total time" $OUT/ba_fr1xyz_every$ev.log | cut -c1-220
done; done
tail -2 $OUT/ba_fr1xyz_every1.log | head -1
gbp_poplar_amd/bin/ba --bal_file data/sequences/fr2robot2.txt --eval_every 100 | grep "Total time" | cut -c1-220
gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt --eval_every 100 | grep "Total time" | cut -c1-220
gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt | grep "Total time" | cut -c1-220
