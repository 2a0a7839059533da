export GIPUMA_HIP_EXPERIMENTS=1
# how many (candidate, view) items the prefilter leaves open in the FIRST half-sweeps (pixel-per-lane refinement there)
sh scripts/gpu_ab.sh <<LIST
C_c0_counts GIPUMA_HIP_COLS_LAUNCHES=0 GIPUMA_HIP_COUNTS=1 GIPUMA_HIP_ET_FORCE=2
C_c0_counts_k16 GIPUMA_HIP_COLS_LAUNCHES=0 GIPUMA_HIP_COUNTS=1 GIPUMA_HIP_ET_FORCE=2 GIPUMA_HIP_LB_K=16
LIST
grep "items/px\|cands/px" gpurun_out/ab/C_c0_counts.err | tail -4 | cut -c1-200
grep "items/px\|cands/px" gpurun_out/ab/C_c0_counts_k16.err | tail -4 | cut -c1-200
