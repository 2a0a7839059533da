#!/bin/sh
# round 6, experiment 5: the frame's last tile column right after the first (tile_of) against the plain column order (bit 21)
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
for rep in 1 2 3; do
for t in 0 $((1<<21)); do
  echo "== GIPUMA_HIP_TUNE=$t"
  GIPUMA_HIP_TUNE=$t python scripts/gpu_r06_time.py C 2>&1 | grep -v amdgpu.ids
done
done
for t in 0 $((1<<21)); do
  echo "== GIPUMA_HIP_TUNE=$t"
  GIPUMA_HIP_TUNE=$t python scripts/gpu_r06_time.py D colour box19 B 2>&1 | grep -v amdgpu.ids
done
