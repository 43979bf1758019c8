cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -o small -- $GRAFT_REPO_ROOT/gbp_poplar_amd/bin/ba --bal_file $GRAFT_REPO_ROOT/data/sequences/fr1xyz.txt --n_iters 600 --eval_every 600 > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_small/small_kernel_stats.csv | cut -c1-150
