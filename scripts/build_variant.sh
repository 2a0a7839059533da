#!/bin/sh
# build a differently configured library for A/B runs (loaded through GIPUMA_HIP_LIB under GIPUMA_HIP_EXPERIMENTS=1):
#   sh scripts/build_variant.sh <name> [-DFLAG ...]   ->  gipuma_amd/csrc/variants/libgipuma_hip_<name>.so
# All three translation units (default flavour, GIPUMA_HIP_FLAG_FAST flavour, GIPUMA_HIP_FLAG_LITERAL flavour) get the extra flags.
# Objects carry the process id in their names (concurrent builds of the same variant do not clobber each other) and a
# failed compile fails the script instead of linking whatever an earlier run left behind.
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$R/gipuma_amd/csrc/variants"
cd "$R/gipuma_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall"
O=variants/$NAME.$$
rm -f $O.exact.o $O.fast.o $O.literal.o
/opt/rocm/bin/hipcc $F "$@" -c -o $O.exact.o gipuma_hip.hip & P1=$!
/opt/rocm/bin/hipcc $F "$@" -c -o $O.fast.o gipuma_hip_fast.hip & P2=$!
/opt/rocm/bin/hipcc $F "$@" -c -o $O.literal.o gipuma_hip_literal.hip & P3=$!
RC=0
wait $P1 || RC=1
wait $P2 || RC=1
wait $P3 || RC=1
if [ $RC -eq 0 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgipuma_hip_$NAME.so $O.exact.o $O.fast.o $O.literal.o || RC=1
fi
rm -f $O.exact.o $O.fast.o $O.literal.o
exit $RC
