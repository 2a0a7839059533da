#!/bin/sh
# round 6, experiment 4: per-workgroup clocks of the fused launches (variant wgt = -DPM_WG_TICKS): how full the GPU's
# workgroup slots are over a launch, and what other dispatch orders would give (scripts/exp/wg_ticks_analyse.py)
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
export GIPUMA_HIP_LIB=$PWD/gipuma_amd/csrc/variants/libgipuma_hip_wgt.so
for w in C; do
  rm -f /tmp/wg_ticks.bin
  GIPUMA_HIP_WG_TICKS=/tmp/wg_ticks.bin python scripts/gpu_r06_time.py $w 2>&1 | grep -v amdgpu.ids
  python scripts/exp/wg_ticks_analyse.py /tmp/wg_ticks.bin
done
