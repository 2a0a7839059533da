#!/bin/sh
# round 6, experiment 3: workgroup -> tile mappings (GIPUMA_HIP_TUNE bit 21: an XCD's tiles in bands spread over the frame,
# band height in bits 8..17; bit 2: no XCD-aware mapping at all) on config C
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
for rep in 1 2; do
for t in 0 $((1<<21)) $(( (1<<21) | (1<<8) )) $(( (1<<21) | (2<<8) )) $(( (1<<21) | (5<<8) )) $(( (1<<21) | (10<<8) )) 4; do
  echo "== GIPUMA_HIP_TUNE=$t"
  GIPUMA_HIP_TUNE=$t python scripts/gpu_r06_time.py C 2>&1 | grep -v amdgpu.ids
done
done
for t in 0 $((1<<21)) $(( (1<<21) | (2<<8) )); do
  echo "== GIPUMA_HIP_TUNE=$t"
  GIPUMA_HIP_TUNE=$t python scripts/gpu_r06_time.py D colour box19 B 2>&1 | grep -v amdgpu.ids
done
