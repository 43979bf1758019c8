#!/bin/bash
# Kernel-level profile of the N>1 code path on ONE GPU (1-rank RCCL group), on the per-GPU shard shape of BASELINE
# config 5 (8 000 cameras x 125 000 landmarks x 1.25 M factors):   bash profiles/run_profile_sharded.sh r01
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_sharded_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/bench.py --steps 110 --warmup 20 --cpu-seconds 0 --profile-steps 0 --force-sharded --cams 8000 --lmks 125000 > $OUT/bench.json 2> $OUT/stats.log
head -12 $OUT/stats/stats_kernel_stats.csv | cut -c1-160
