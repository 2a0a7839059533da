#!/bin/sh
# round 6, experiment 6: the library built with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 (scalar spills go to scratch memory, not
# into vector lanes) against the default build: time per view of the bench configurations and modes
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
for rep in 1 2; do
  python scripts/gpu_r06_time.py C C@fast C@literal 2>&1 | grep -v amdgpu.ids
  GIPUMA_HIP_LIB=$V/libgipuma_hip_nosv.so python scripts/gpu_r06_time.py C C@fast C@literal 2>&1 | grep -v amdgpu.ids
done
python scripts/gpu_r06_time.py D colour box19 B 2>&1 | grep -v amdgpu.ids
GIPUMA_HIP_LIB=$V/libgipuma_hip_nosv.so python scripts/gpu_r06_time.py D colour box19 B 2>&1 | grep -v amdgpu.ids
