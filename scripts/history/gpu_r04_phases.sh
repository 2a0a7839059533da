#!/bin/sh
# round 4: where a fused plane-keyed launch spends its time (phase clocks), config C and D; config D's kernel series
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
sh scripts/gpu_ab.sh <<LIST
C_default
C_counts GIPUMA_HIP_COUNTS=1
C_default2
LIST
grep "batches\|phase ticks" $R/gpurun_out/ab/C_counts.err | tail -2
echo "--- config D"
sh scripts/gpu_ab.sh --config D <<LIST
D_default
D_counts GIPUMA_HIP_COUNTS=1
LIST
grep "batches\|phase ticks" $R/gpurun_out/ab/D_counts.err | tail -2
BENCH_ARGS="--config D" sh scripts/gpu_prof.sh r04_D_series | grep -v "at::\|Cijk" | head -60
