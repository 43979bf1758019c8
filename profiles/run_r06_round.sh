#!/bin/bash
# Round-6 record run (GPU box, repo root):  bash profiles/run_r06_round.sh   -> gpurun_out/round_r06/  (profiles/r06_* are made from it)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round_r06
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 50 --warmup 10 --pmc off --cpu-seconds 0 --small-configs off --windows 0 --sustained-seconds 0 > /dev/null 2>&1   # warm the box
# 1. the driver's command under rocprofv3 --kernel-trace --stats, WITHOUT the PMC children (no profiler inside a profiled process)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver_trace -o drv -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --pmc off > $OUT/bench_driver_under_rocprof.json 2> $OUT/bench_driver_under_rocprof.err )
# 2. the same command plain (what BENCH_r06 will hold), twice — the PMC traffic comes from these; and the default invocation
python3 bench.py --gpus 1 --steps 20 --warmup 5 --save-traffic $OUT/traffic_S1.json > $OUT/bench_driver_1.json 2> /dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_2.json 2> /dev/null
python3 bench.py > $OUT/bench_default.json 2> /dev/null
# 3. the config-5 shard shape: plain ctx, and the line an N > 1 run prints (1-rank communicator)
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --small-configs off --cams 8000 --lmks 125000 > $OUT/bench_c5shape_plain.json 2> /dev/null
python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 2> /dev/null | grep '^{' > $OUT/bench_c5shape_driverline.json
python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 200 --warmup 20 --cpu-seconds 0 --pmc off 2> /dev/null | grep '^{' > $OUT/bench_c5shape_native_200.json
# 4. real ranks sharing the one GPU: the N > 1 code path with N processes (config-5 family, host-staged transport)
for n in 2 4 8; do python3 bench.py --gpus $n --share-gpu --steps 20 --warmup 5 --cpu-seconds 0 --pmc off 2> $OUT/bench_share_$n.err | grep '^{' > $OUT/bench_share_$n.json; done
# 5. configs 1-3 through the CLIs, default loop and --eval_every 100 (second of two runs kept), outputs hashed, start-up attributed
for seq in fr1xyz fr2robot2 fr1desk; do for ev in 1 100; do for rep in 1 2; do
  gbp_poplar_amd/bin/ba --bal_file data/sequences/$seq.txt --eval_every $ev > $OUT/ba_${seq}_every$ev.log 2>&1; done; done; done
for ev in 1 100; do for rep in 1 2; do gbp_poplar_amd/bin/slam --bal_file data/sequences/fr2robot2.txt --eval_every $ev > $OUT/slam_fr2robot2_every$ev.log 2>&1; done; done
for f in $OUT/ba_*_every1.log $OUT/slam_fr2robot2_every1.log; do echo "$(basename $f) $(grep -v 'Total time' $f | md5sum | cut -c1-32)"; done > $OUT/cli_md5.txt
python3 profiles/time_cli.py 5 0.5 > $OUT/cli_startup.txt 2>&1      # 0.5 s idle in front of every run (r06_cli_pause.txt)
python3 profiles/big_file_cli.py 4 > $OUT/cli_bigfile.txt 2>&1
# 6. the reference's default loop on the 1M-factor graph THROUGH bin/ba
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from gbp_poplar_amd import hostlib
hostlib.bal_write("/tmp/s1.txt", hostlib.synth_generate(1000, 100000, 10, 20200303))
PY
for ev in 1 100; do for rep in 1 2; do gbp_poplar_amd/bin/ba --bal_file /tmp/s1.txt --n_iters 300 --eval_every $ev > $OUT/ba_S1_every$ev.log 2>&1; done; done
for s in fr1xyz fr2robot2 fr1desk; do python3 profiles/time_bursts.py $s 10 2>/dev/null | tail -1; done > $OUT/persist_bursts.txt
for f in $OUT/ba_*.log $OUT/slam_*.log; do echo "== $(basename $f)"; grep -E "Total time" $f | cut -c1-400; grep -E "^Iter" $f | tail -1; done > $OUT/cli_summary.txt
cp $(find $OUT/driver_trace -name '*kernel_stats.csv' | head -1) $OUT/r06_kernel_stats.csv
