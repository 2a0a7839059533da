#!/bin/sh
# round 4: config D's refinement stage -- bounded evaluation forced on earlier, prefilter lengths
export GIPUMA_HIP_EXPERIMENTS=1
sh scripts/gpu_ab.sh --config D <<LIST
D_default
D_force2 GIPUMA_HIP_ET_FORCE=2
D_force2_k32 GIPUMA_HIP_ET_FORCE=2 GIPUMA_HIP_LB_K=32
D_force2_k16 GIPUMA_HIP_ET_FORCE=2 GIPUMA_HIP_LB_K=16
D_k32 GIPUMA_HIP_LB_K=32
D_lboff GIPUMA_HIP_LB_K=-1
D_counts GIPUMA_HIP_COUNTS=1
LIST
grep "items/px\|cands/px" gpurun_out/ab/D_counts.err | tail -4 | cut -c1-220
