#!/bin/bash
# Round-4 record run (GPU box, repo root):  bash profiles/run_r04_round.sh  -> gpurun_out/round_r04/ (summaries are copied into profiles/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash profiles/run_round.sh r04 > gpurun_out/round_r04.log 2>&1
OUT=$REPO/gpurun_out/round_r04
mkdir -p $OUT
# per-dispatch durations + FETCH / WRITE of ordinary vs lock-step k_sweep dispatches (sweeps 0..35 of the ./ba flow on S1)
cd /tmp && export TMPDIR=/tmp
CHILD="$REPO/bench.py --pmc-child --steps 24 --warmup 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ld_trace -o t -- python3 $CHILD > /dev/null 2> $OUT/ld_trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ld_fetch -o f -- python3 $CHILD > /dev/null 2> $OUT/ld_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ld_write -o w -- python3 $CHILD > /dev/null 2> $OUT/ld_write.log
cd $REPO
python3 profiles/relin_dispatches.py $OUT/ld_trace $OUT/ld_fetch $OUT/ld_write > gpurun_out/r04_relin_dispatches.csv 2> gpurun_out/r04_relin_dispatches.txt
bash profiles/prof_small.sh r04 > gpurun_out/prof_small_r04.log 2>&1
python3 profiles/persist_trace.py fr1xyz > gpurun_out/r04_persist_trace_fr1xyz.txt 2>&1
python3 profiles/persist_trace.py fr1xyz each > gpurun_out/r04_persist_trace_fr1xyz_each.txt 2>&1
python3 profiles/persist_trace.py fr2robot2 > gpurun_out/r04_persist_trace_fr2robot2.txt 2>&1
bash profiles/run_pmc_detail.sh r04 > gpurun_out/pmc_detail_r04.log 2>&1
