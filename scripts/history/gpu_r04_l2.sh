#!/bin/sh
# round 4: box 25 chains -- two lanes per task for batches of up to 32 tasks, byte-indexed weights otherwise
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed or kernel_variants" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
V=$R/gipuma_amd/csrc/variants
sh scripts/gpu_ab.sh --config D <<LIST
D_prev GIPUMA_HIP_LIB=$V/libgipuma_hip_prev.so
D_new
D_new_counts GIPUMA_HIP_COUNTS=1
LIST
grep "batches\|phase ticks" $R/gpurun_out/ab/D_new_counts.err | tail -2
sh scripts/gpu_ab.sh <<LIST
C_new
LIST
