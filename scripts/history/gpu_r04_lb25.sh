#!/bin/sh
# round 4: box 25's streamed prefilter walk without register rotation (two sets): parity subset + A/B on config D
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/gipuma_amd/csrc/variants
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -k "lower_bound or config_d_every_launch or (plane_keyed and 25)" 2>&1 | tail -n 2
sh scripts/gpu_ab.sh --config D <<LIST
D_base GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
D_new
D_base2 GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
D_new2
LIST
