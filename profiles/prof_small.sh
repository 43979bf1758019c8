#!/bin/bash
# Kernel-level profile of a small real sequence through the CLI (run on the GPU box from the repo root):
#   bash profiles/prof_small.sh [fr1xyz]      -> gpurun_out/prof_small_<seq>_kernel_stats.csv
SEQ=${1:-fr1xyz}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_small_$SEQ -o small -- $REPO/gbp_poplar_amd/bin/ba --bal_file $REPO/data/sequences/$SEQ.txt --n_iters 600 > /dev/null 2>&1
cp $REPO/gpurun_out/prof_small_$SEQ/small_kernel_stats.csv $REPO/gpurun_out/prof_small_${SEQ}_kernel_stats.csv
cut -d, -f1-4,7 $REPO/gpurun_out/prof_small_${SEQ}_kernel_stats.csv | head -8
