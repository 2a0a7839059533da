#!/bin/sh
# build a differently configured library for A/B runs (loaded through GIPUMA_HIP_LIB under GIPUMA_HIP_EXPERIMENTS=1):
#   sh scripts/build_variant.sh <name> [-DFLAG ...]   ->  gipuma_amd/csrc/variants/libgipuma_hip_<name>.so
# All three translation units (exact flavour, GIPUMA_HIP_FLAG_FAST flavour, GIPUMA_HIP_FLAG_LITERAL flavour) get the extra flags.
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/gipuma_amd/csrc/variants
cd $R/gipuma_amd/csrc || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall"
/opt/rocm/bin/hipcc $F "$@" -c -o variants/$NAME.exact.o gipuma_hip.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o variants/$NAME.fast.o gipuma_hip_fast.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o variants/$NAME.literal.o gipuma_hip_literal.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgipuma_hip_$NAME.so variants/$NAME.exact.o variants/$NAME.fast.o variants/$NAME.literal.o \
  && rm -f variants/$NAME.exact.o variants/$NAME.fast.o variants/$NAME.literal.o
