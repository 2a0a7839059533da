export GIPUMA_HIP_EXPERIMENTS=1
V=$PWD/gipuma_amd/csrc/variants
sh scripts/gpu_ab.sh --config D <<LIST
D_prev GIPUMA_HIP_LIB=$V/libgipuma_hip_prev.so
D_l01 GIPUMA_HIP_LIB=$V/libgipuma_hip_laps0x01.so
D_l10 GIPUMA_HIP_LIB=$V/libgipuma_hip_laps0x10.so
D_l20 GIPUMA_HIP_LIB=$V/libgipuma_hip_laps0x20.so
D_l40 GIPUMA_HIP_LIB=$V/libgipuma_hip_laps0x40.so
LIST
