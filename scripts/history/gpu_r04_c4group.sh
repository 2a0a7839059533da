#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
sh scripts/gpu_ab.sh --colour <<LIST
c4_nogroup GIPUMA_HIP_GROUP_FROM=-1
c4_fused
c4_unfused GIPUMA_HIP_GROUP_FUSED=0
c4_fused_g4 GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_PUSH_LAUNCHES=4
c4_fused_g8 GIPUMA_HIP_GROUP_FROM=8 GIPUMA_HIP_PUSH_LAUNCHES=8
LIST
echo "--- C (odd rows) and D"
sh scripts/gpu_ab.sh <<LIST
C_default
C_default2
LIST
sh scripts/gpu_ab.sh --config D <<LIST
D_default
LIST
