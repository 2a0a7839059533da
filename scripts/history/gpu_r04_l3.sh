#!/bin/sh
# round 4: plane-keyed strips: one loop with a scalar branch on the reciprocal instead of two specialised loops (vmcnt waits)
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
V=$R/gipuma_amd/csrc/variants
sh scripts/gpu_ab.sh <<LIST
C_base GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
C_new
C_base2 GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
C_new2
C_counts GIPUMA_HIP_COUNTS=1
LIST
sh scripts/gpu_ab.sh --config D <<LIST
D_base GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
D_new
D_counts GIPUMA_HIP_COUNTS=1
LIST
sh scripts/gpu_ab.sh --colour <<LIST
col_base GIPUMA_HIP_LIB=$V/libgipuma_hip_base.so
col_new
LIST
grep "batches" gpurun_out/ab/C_counts.err gpurun_out/ab/D_counts.err | tail -2
