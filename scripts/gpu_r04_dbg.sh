export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
export GIPUMA_HIP_LIB=$V/libgipuma_hip_laps0b.so
run() { echo "--- $*"; env "$@" timeout 300 python scripts/exp/colour_fused_diff.py 1 2>&1 | grep "iterations 1 run 0"; }
run X=1
run GIPUMA_HIP_TUNE=4194304
run GIPUMA_HIP_TUNE=8388608
run GIPUMA_HIP_TUNE=33554432
run GIPUMA_HIP_TUNE=64
run GIPUMA_HIP_LB_K=-1
run GIPUMA_HIP_ET_FORCE=0
run GIPUMA_HIP_TUNE=46137408
