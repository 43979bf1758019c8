#!/bin/bash
TAG=${1:-r04g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_experiments.py tests/test_cli.py -m gpu -q -x -k "persistent or iterate_eval or health or config1 or config3 or cli or two_processes or ipus or time_out" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
VARIANTS="default nometroles" bash profiles/run_r04d.sh $TAG
cat $OUT/run.log
timeout 300 python3 profiles/persist_trace.py fr1xyz each > $OUT/persist_trace_fr1xyz_each.txt 2>&1; head -24 $OUT/persist_trace_fr1xyz_each.txt
