#!/bin/bash
# Round-6 (GPU box, repo root):  bash profiles/run_r06_timeline.sh <tag>  -> gpurun_out/r06_timeline_<tag>/
# Kernel-level timeline of the sharded iteration on the config-5 shard shape (1-rank RCCL communicator), one- and two-stream
# schedule, beside the plain ctx on the same graph; then the untraced lines of the driver's command.  profiles/sharded_timeline.py
# turns the traces into profiles/r06_sharded_timeline.md.
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_timeline_$TAG
mkdir -p $OUT
cd $REPO
for ss in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && GBP_COMM_SINGLE_STREAM=$ss rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_ss$ss -o t -- \
      python3 $REPO/bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 40 --warmup 10 --cpu-seconds 0 --pmc off --preflight 0 \
      --profile-steps 0 --windows 0 --sustained-seconds 0 > $OUT/bench_trace_ss$ss.json 2> $OUT/bench_trace_ss$ss.err )
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_plain -o t -- \
    python3 $REPO/bench.py --gpus 1 --cams 8000 --lmks 125000 --steps 40 --warmup 10 --cpu-seconds 0 --pmc off --small-configs off \
    --profile-steps 0 --windows 0 --sustained-seconds 0 > $OUT/bench_trace_plain.json 2> $OUT/bench_trace_plain.err )
for rep in 1 2; do
python3 bench.py --gpus 1 --force-sharded --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off 2> /dev/null | grep '^{' > $OUT/bench_c5shape_sharded_$rep.json
python3 bench.py --gpus 1 --cams 8000 --lmks 125000 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off --small-configs off 2> /dev/null | grep '^{' > $OUT/bench_c5shape_plain_$rep.json
done
python3 profiles/sharded_timeline.py "sharded, one stream=$OUT/trace_ss1" "sharded, two streams=$OUT/trace_ss0" "plain ctx, hipGraph=$OUT/trace_plain" > $OUT/timeline.md
