#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/run_profile.sh r01
# Pass 1: rocprofv3 --kernel-trace --stats of the default bench workload (S1, 1M factors).
# Pass 2/3: PMC counters FETCH_SIZE and WRITE_SIZE in their own runs (MI355X_MICROARCH.md, HBM section:
#           TCC slots do not fit both in one pass; FETCH_SIZE reads 1/2 of wide coalesced streams on gfx950).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/bench.py --steps 100 --warmup 20 --cpu-seconds 0 > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py --steps 5 --warmup 12 --cpu-seconds 0 --profile-steps 0 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py --steps 5 --warmup 12 --cpu-seconds 0 --profile-steps 0 > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
python3 $REPO/profiles/summarise.py $OUT $REPO/gpurun_out/profile_summary_$TAG.md
