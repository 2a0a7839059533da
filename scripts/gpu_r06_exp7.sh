#!/bin/sh
# round 6, experiment 7: the fused launches' dispatch order from the same colour's previous workgroup durations
# (pm::tile_order_kernel, GIPUMA_HIP_TILE_ORDER=1) against the plain order; parity of the whole frame first
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
GIPUMA_HIP_TILE_ORDER=1 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "config_c_whole_frame or plane_keyed_propagation_is_bit or fused_kernel_equals" 2>&1 | tail -3
for rep in 1 2 3; do
for t in 0 1; do
  echo "== GIPUMA_HIP_TILE_ORDER=$t"
  GIPUMA_HIP_TILE_ORDER=$t python scripts/gpu_r06_time.py C 2>&1 | grep -v amdgpu.ids
done
done
for t in 0 1; do
  echo "== GIPUMA_HIP_TILE_ORDER=$t"
  GIPUMA_HIP_TILE_ORDER=$t python scripts/gpu_r06_time.py D box19 C@fast C@literal 2>&1 | grep -v amdgpu.ids
done
