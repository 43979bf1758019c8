#!/bin/bash
TAG=${1:-r04f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_device_math.py tests/test_gpu_parity.py -m gpu -q -x -k "so3exp or config1 or other_sequences or config3 or persistent or lockstep or relinearising or golden" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
VARIANTS="default sincossep" bash profiles/run_r04d.sh $TAG
cat $OUT/run.log
