#!/bin/sh
# Compile the reference's own device code for the CPU:  sh build_ref.sh <reference-dir> <out.so>
# The part of gipuma.cu before its host launcher `void gipuma(GlobalState&)` (which needs nvcc's
# <<<...>>>) is piped, untouched, between the CUDA-on-CPU shim and the harness.  Nothing of the
# reference is written to disk except the resulting binary.  (-fno-extern-tls-init: the block-scope `extern __shared__`
# tile of the kernels is a thread_local array without a dynamic initialiser; g++ would call a TLS wrapper it never emits.)
set -e
REF=$1
OUT=$2
HERE=$(cd "$(dirname "$0")" && pwd)
LAUNCHER=$(grep -n '^void gipuma(GlobalState &gs)' "$REF/gipuma.cu" | head -1 | cut -d: -f1)
[ -n "$LAUNCHER" ] || { echo "build_ref: launcher not found in $REF/gipuma.cu" >&2; exit 1; }
LAST=$((LAUNCHER - 2))   # drop the `template< typename T >` line in front of it too
mkdir -p "$(dirname "$OUT")"
{
  echo '#include "ref_cuda_on_cpu.h"'
  echo '#line 1 "reference/gipuma.cu"'
  sed -n "1,${LAST}p" "$REF/gipuma.cu"
  echo "#line 1 \"$HERE/ref_harness.cpp\""
  cat "$HERE/ref_harness.cpp"
} | g++ -x c++ -std=gnu++14 -O2 -fPIC -shared -fopenmp -fno-extern-tls-init -ffp-contract=off -mavx2 -mfma -fno-math-errno -w \
        -I"$HERE/cuda" -I"$HERE" -I"$REF" -o "$OUT" -
echo "built $OUT from $REF/gipuma.cu lines 1..$LAST"
