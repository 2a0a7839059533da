#!/bin/sh
# Compile the reference-signature adapter against the reference's own headers:
#   sh build_adapter.sh <reference-dir>   ->  gipuma_amd/csrc/adapter/libgipuma_runcuda.so
# (only possible where the reference tree exists; the .so travels to the GPU box)
set -e
REF=$1
HERE=$(cd "$(dirname "$0")" && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -include: the reference's managed.h pulls "helper_cuda.h" before any CUDA header, so the compat
# header has to come first (it also keeps NVIDIA's vendored helper_cuda.h out)
$HIPCC -O2 -std=c++17 -fPIC -shared -w -D__HIP_PLATFORM_AMD__ \
    -include "$HERE/cuda_compat/gipuma_cuda_compat.h" -I"$HERE/cuda_compat" -I"$REF" \
    -o "$HERE/libgipuma_runcuda.so" "$HERE/runcuda.cpp" "$HERE/adapter_selftest.cpp" \
    -L"$HERE/.." -lgipuma_hip -Wl,-rpath,'$ORIGIN/..'
echo "built $HERE/libgipuma_runcuda.so against $REF"
