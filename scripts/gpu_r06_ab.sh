#!/bin/sh
# A/B of one experiment switch on config C: sh scripts/gpu_r06_ab.sh <tune bits of arm B> [mode ...]
# prints ms per view and the half-sweep series of both arms (GIPUMA_HIP_TUNE under GIPUMA_HIP_EXPERIMENTS=1)
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
TUNE=$1
for arm in 0 $TUNE 0 $TUNE; do
  echo "== GIPUMA_HIP_TUNE=$arm"
  GIPUMA_HIP_TUNE=$arm python scripts/gpu_r06_first.py time 2>&1 | grep "config C"
done
