#!/bin/sh
# fused against two launches, FAST flavour, free-running, for a list of variant libraries
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  echo "== $v"
  GIPUMA_HIP_LIB=$PWD/gipuma_amd/csrc/variants/libgipuma_hip_$v.so python scripts/exp/checked_fast_probe3.py fast 2>&1 | grep -v "amdgpu\|CHECKED" | grep "after 5" | cut -c1-140
done
