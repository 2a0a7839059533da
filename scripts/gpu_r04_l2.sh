#!/bin/sh
# round 4: box 25 with two lanes per task (float weights), global-address-space pointers in the problem block
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed or kernel_variants or push_propagation or seen_rule" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
V=$R/gipuma_amd/csrc/variants
sh scripts/gpu_ab.sh --config D <<LIST
D_prev GIPUMA_HIP_LIB=$V/libgipuma_hip_prev.so
D_new
LIST
sh scripts/gpu_ab.sh <<LIST
C_new
LIST
