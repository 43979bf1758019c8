#!/bin/bash
# Round-6 item 1 (GPU box, repo root):  bash profiles/run_r06_world8.sh   -> gpurun_out/r06_world8/
#   a. the world the driver will start, as eight REAL processes on the one GPU: bench.py --gpus 8 --share-gpu (config 5 itself:
#      8 000 cameras x 1 000 000 landmarks x 10 000 000 factors, host-staged transport), and bin/ba --ipus 8 on a config-5-shaped
#      file (8 000 x 250 000 x 2.5 M factors: the text file of the full graph is 0.7 GB, parsed by each of the 8 ranks)
#   b. the kernel-level timeline of ONE sharded iteration on the config-5 shard shape (--force-sharded, 1-rank RCCL communicator):
#      rocprofv3 --kernel-trace --memory-copy-trace of a short run; profiles/sharded_timeline.py turns the CSVs into the table
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_world8
mkdir -p $OUT
cd $REPO
nproc > $OUT/host.txt; free -g >> $OUT/host.txt
# a1. bench, eight ranks sharing the GPU
( time timeout 1500 python3 bench.py --gpus 8 --share-gpu --steps 20 --warmup 5 --cpu-seconds 0 --pmc off ) > $OUT/bench_share_8.json 2> $OUT/bench_share_8.err
# a2. bin/ba --ipus 8 on a config-5-shaped file + the plain single-process run of the same file
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from gbp_poplar_amd import hostlib
hostlib.bal_write("/tmp/c5.txt", hostlib.synth_generate(8000, 250000, 10, 20200303))
PY
ls -la /tmp/c5.txt >> $OUT/host.txt
( time timeout 900 gbp_poplar_amd/bin/ba --bal_file /tmp/c5.txt --n_iters 40 --eval_every 10 --ipus 8 ) > $OUT/ba_ipus8_c5.log 2>&1
( time timeout 900 gbp_poplar_amd/bin/ba --bal_file /tmp/c5.txt --n_iters 40 --eval_every 10 ) > $OUT/ba_ipus1_c5.log 2>&1
# b. timeline of the sharded iteration: one-stream and two-stream schedule, 1-rank communicator, direct launches
for ss in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && GBP_COMM_SINGLE_STREAM=$ss rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_ss$ss -o t -- \
      python3 $REPO/bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 40 --warmup 10 --cpu-seconds 0 --pmc off --preflight 0 \
      --profile-steps 0 --windows 0 --sustained-seconds 0 > $OUT/bench_trace_ss$ss.json 2> $OUT/bench_trace_ss$ss.err )
done
# the same shape through the plain ctx (what the sharded line is compared with)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_plain -o t -- \
    python3 $REPO/bench.py --gpus 1 --cams 8000 --lmks 125000 --steps 40 --warmup 10 --cpu-seconds 0 --pmc off --small-configs off \
    --profile-steps 0 --windows 0 --sustained-seconds 0 > $OUT/bench_trace_plain.json 2> $OUT/bench_trace_plain.err )
# untraced reference lines of the three (driver's command)
python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off 2> /dev/null | grep '^{' > $OUT/bench_c5shape_sharded.json
python3 bench.py --gpus 1 --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --small-configs off 2> /dev/null | grep '^{' > $OUT/bench_c5shape_plain.json
find $OUT -name '*.csv' | xargs ls -la >> $OUT/host.txt
du -sh $OUT >> $OUT/host.txt
