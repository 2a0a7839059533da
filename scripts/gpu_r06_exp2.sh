#!/bin/sh
# round 6, experiment 2: the unrolled generic push stencil as the default (parity of the push / plane-keyed tests, times of box 19
# with its new schedule, config D), the colour push stencil unrolled (variant c4u), and the phase clocks of the fused kernel
# (GIPUMA_HIP_COUNTS: how busy the workgroup slots are over a launch)
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "push_propagation or maximum_number_of_views_pushed or plane_keyed_propagation_is_bit" 2>&1 | tail -3
python scripts/gpu_r06_time.py box19 D C
python scripts/gpu_r06_time.py colour
GIPUMA_HIP_LIB=$V/libgipuma_hip_c4u.so python scripts/gpu_r06_time.py colour
echo "== phase clocks"
GIPUMA_HIP_COUNTS=1 python scripts/gpu_r06_time.py C 2>&1 | grep -v amdgpu.ids | tail -12
