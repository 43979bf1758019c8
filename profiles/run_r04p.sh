#!/bin/bash
TAG=${1:-r04p}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
exec > $OUT/run.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_experiments.py -m gpu -q -x -k "persistent or iterate_eval or health or config1 or config3 or time_out or captured" 2>&1 | tail -3
for round in 1 2 3; do
  for v in default prev; do
    if [ $v = default ]; then L=$REPO/gbp_poplar_amd/libgbp_mi355x.so; else L=$REPO/profiles/_bin/$v/libgbp_mi355x.so; fi
    for seq in fr1xyz fr2robot2; do echo -n "$v: "; GBP_LIB=$L python3 profiles/time_bursts.py $seq 10 2>/dev/null; done
  done
done
