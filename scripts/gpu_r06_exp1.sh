#!/bin/sh
# round 6, experiment 1: the generic push stencil loop fully unrolled (variants pfu, pfu4) on boxes 19 / 25, and the schedules
# of box 19, config D and colour re-tuned under the round-6 default model
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
for w in box19 D; do
  python scripts/gpu_r06_sched.py $w - 2,2,2 3,3,3 4,4,4
  for v in pfu pfu4; do GIPUMA_HIP_LIB=$V/libgipuma_hip_$v.so python scripts/gpu_r06_sched.py $w - 2,2,2 3,3,3 4,4,4; done
done
python scripts/gpu_r06_sched.py colour - 2,4,3 3,4,3 4,4,4 3,3,3 3,4,4 2,2,2
python scripts/gpu_r06_time.py A B C
