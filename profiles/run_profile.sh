#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun from the repo root):   bash profiles/run_profile.sh r02
# Pass 1: rocprofv3 --kernel-trace --stats of the default bench workload (S1, 1M factors), the driver's command line.
# Pass 2: the bench's own live PMC passes (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 runs, MI355X_MICROARCH.md HBM:
#         the TCC slots do not hold both; FETCH_SIZE reads 1/2 of wide coalesced streams on gfx950), CSVs kept.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pmc off > $OUT/stats_bench.json 2> $OUT/stats.log
cd $REPO
python3 bench.py --steps 20 --warmup 5 --keep-pmc $OUT/pmc --save-traffic $OUT/traffic_S1.json > $OUT/bench.json 2> $OUT/bench.err
python3 $REPO/profiles/summarise.py $OUT $REPO/gpurun_out/profile_summary_$TAG.md
cp $OUT/stats/stats_kernel_stats.csv $REPO/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
