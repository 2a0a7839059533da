#!/bin/sh
# round 4: dis_fold + bigger sample buffers on config C; plane-keyed propagation on config D (box 25)
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed or kernel_variants or lower_bound or two_phase or push_propagation or colour_full or config_a" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
sh scripts/gpu_ab.sh <<LIST
C_default
C_counts GIPUMA_HIP_COUNTS=1
C_default2
LIST
grep "batches" $R/gpurun_out/ab/C_counts.err | tail -1
echo "--- config D"
sh scripts/gpu_ab.sh --config D <<LIST
D_nogroup GIPUMA_HIP_GROUP_FROM=-1
D_fused
D_unfused GIPUMA_HIP_GROUP_FUSED=0
D_fused_g4 GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_PUSH_LAUNCHES=4
D_counts GIPUMA_HIP_COUNTS=1
LIST
grep "batches" $R/gpurun_out/ab/D_counts.err | tail -1
echo "--- colour"
sh scripts/gpu_ab.sh --colour <<LIST
colour_default
LIST
