#!/bin/sh
# time per config-C view of the default library and of every library under gipuma_amd/csrc/variants/ named on the command
# line (A/B of compile-time choices): sh scripts/gpu_r06_variants.sh b32 pd2 ...
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
for round in 1 2; do
  echo "== default"; python scripts/gpu_r06_first.py time 2>&1 | grep "config C default"
  for v in "$@"; do
    echo "== $v"; GIPUMA_HIP_LIB=$PWD/gipuma_amd/csrc/variants/libgipuma_hip_$v.so python scripts/gpu_r06_first.py time 2>&1 | grep "config C default"
  done
done
