#!/bin/bash
# Kernel-level profile of the small real sequences through the CLIs (run on the GPU box from the repo root):
#   bash profiles/prof_small.sh [tag]      -> gpurun_out/<tag>_small_<run>_kernel_stats.csv (copy into profiles/)
# Runs: ba fr1xyz with --eval_every 100 (bursts: k_persist), with the default per-iteration metric (gbp_iterate_eval_each:
# k_persist with the metric phases), the same two with GBP_PERSIST=-1 (k_sweep + k_beliefs [+ k_means + k_eval]), and slam.
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_small_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, binary, args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o p -- "$@" > $OUT/$name.log 2>&1
  cp $OUT/$name/p_kernel_stats.csv $REPO/gpurun_out/${TAG}_small_${name}_kernel_stats.csv
  echo "== $name"; cut -d, -f1-4 $REPO/gpurun_out/${TAG}_small_${name}_kernel_stats.csv | cut -c1-110 | head -7
}
BA=$REPO/gbp_poplar_amd/bin/ba; SLAM=$REPO/gbp_poplar_amd/bin/slam; SEQ=$REPO/data/sequences
prof fr1xyz_every100 $BA --bal_file $SEQ/fr1xyz.txt --eval_every 100
prof fr1xyz_default $BA --bal_file $SEQ/fr1xyz.txt
export GBP_PERSIST=-1
prof fr1xyz_every100_twokernels $BA --bal_file $SEQ/fr1xyz.txt --eval_every 100
prof fr1xyz_default_twokernels $BA --bal_file $SEQ/fr1xyz.txt
unset GBP_PERSIST
prof slam_fr2robot2_default $SLAM --bal_file $SEQ/fr2robot2.txt
prof slam_fr2robot2_every100 $SLAM --bal_file $SEQ/fr2robot2.txt --eval_every 100
