#!/bin/bash
# A product-library variant for A/B measurements:  bash profiles/build_variant.sh <name> <extra hipcc flags...>
#   -> profiles/_bin/<name>/libgbp_mi355x.so   (run a CLI against it with LD_LIBRARY_PATH=profiles/_bin/<name>)
set -e
NAME=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
mkdir -p $REPO/profiles/_bin/$NAME
C=$REPO/gbp_poplar_amd/csrc
hipcc -shared -o $REPO/profiles/_bin/$NAME/libgbp_mi355x.so -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function "$@" \
  -x hip $C/gbp_kernels.hip $C/gbp_api_ctx.cpp $C/gbp_api_launch.cpp $C/gbp_api_persist.cpp $C/gbp_api_eval.cpp $C/gbp_api_comm.cpp $C/gbp_api_debug.cpp $C/gbp_layout.cpp $C/gbp_comm.cpp $C/gbp_host.cpp -ldl -pthread -Wl,--version-script=$C/gbp_exports.map
echo built profiles/_bin/$NAME
