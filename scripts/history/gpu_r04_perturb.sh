#!/bin/sh
# round 4: the gray fused plane-keyed kernels under compile perturbations (generic pointers, profiling hooks compiled out, -O2):
# plane-keyed, kernel-variant and whole-solve parity tests with each library
export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
for lib in generic laps0 O2; do
  echo "--- $lib"
  GIPUMA_HIP_LIB=$V/libgipuma_hip_$lib.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -k "plane_keyed or kernel_variants or stepped or patchy or config_a or config_b" 2>&1 | tail -n 2
done
