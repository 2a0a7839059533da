#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# build a differently configured library for A/B runs (loaded through GIPUMA_HIP_LIB):
#   sh scripts/build_variant.sh <name> [-DFLAG ...]   ->  gipuma_amd/csrc/variants/libgipuma_hip_<name>.so
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/gipuma_amd/csrc/variants
cd $R/gipuma_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
    -fPIC -shared -Wall "$@" -o variants/libgipuma_hip_$NAME.so gipuma_hip.hip
