#!/bin/sh
# Compile the reference's own camera front-end for the CPU:  sh build_hostref.sh <reference-dir> <out-binary>
# getCameraParameters (cameraGeometryUtils.h) and selectViews (main.cpp) with the readers they call are piped,
# untouched, between the CUDA-on-CPU shim, the functional mini OpenCV (opencv2/cv_mini.hpp) and the harness.
# Nothing of the reference is written to disk except the resulting binary (oracle/_ref/, git-ignored).
set -e
REF=$1
OUT=$2
HERE=$(cd "$(dirname "$0")" && pwd)
SHIM=$(cd "$HERE/.." && pwd)
SV0=$(grep -n '^static void selectViews' "$REF/main.cpp" | head -1 | cut -d: -f1)
SV1=$(grep -n '^static void delTexture' "$REF/main.cpp" | head -1 | cut -d: -f1)
RD1=$(grep -n '^static void readKRtFileMiddlebury' "$REF/fileIoUtils.h" | head -1 | cut -d: -f1)
[ -n "$SV0" ] && [ -n "$SV1" ] && [ -n "$RD1" ] || { echo "build_hostref: anchors not found" >&2; exit 1; }
mkdir -p "$(dirname "$OUT")"
{
  echo '#include "ref_cuda_on_cpu.h"'
  echo '#undef expf'
  echo '#include <algorithm>'
  echo '#include <ctime>'
  echo '#include "main.h"'
  echo '#include "algorithmparameters.h"'
  echo '#include "cameraparameters.h"'
  echo '#line 1 "reference/fileIoUtils.h"'
  sed -n "1,$((RD1 - 1))p" "$REF/fileIoUtils.h"
  # (readKRtFileMiddlebury takes its camera vector BY VALUE -- it has no effect in the reference -- and needs
  #  hconcat / stream extraction into matrix elements; the .P path never calls it)
  echo 'static void readKRtFileMiddlebury(const string, vector<Camera>, InputFiles) { abort(); }'
  echo '#include "cameraGeometryUtils.h"'
  echo '#line '"$SV0"' "reference/main.cpp"'
  sed -n "${SV0},$((SV1 - 1))p" "$REF/main.cpp"
  echo "#line 1 \"$HERE/hostref_harness.cpp\""
  cat "$HERE/hostref_harness.cpp"
} | g++ -x c++ -std=gnu++14 -O1 -fopenmp -ffp-contract=off -w \
        -I"$HERE" -I"$SHIM/cuda" -I"$SHIM" -I"$REF" -o "$OUT" -
echo "built $OUT from $REF/{cameraGeometryUtils.h, fileIoUtils.h:1..$((RD1 - 1)), main.cpp:$SV0..$((SV1 - 1))}"
