#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# config-C evidence only: kernel trace + per-launch series, PMC passes, the default bench line
#   sh scripts/evidence_c.sh   ->  gpurun_out/final/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
sh $R/scripts/gpu_prof.sh final_prof GIPUMA_HIP_LAUNCH_TIMES=1 > $O/prof.txt 2>&1
cp $R/gpurun_out/final_prof/kernel_stats.txt $R/gpurun_out/final_prof/series_named.txt $R/gpurun_out/final_prof/err.txt $O/ 2>/dev/null
sh $R/scripts/pmc_passes.sh final_pmc > $O/pmc.txt 2>&1
cp $R/gpurun_out/final_pmc/pmc_summary*.json $O/ 2>/dev/null
# the bench line imports these counters: hand it the ones just collected (the same files are then committed)
cp $O/pmc_summary.json $R/profiles/pmc_latest.json
cp $O/pmc_summary_pixel_per_lane.json $R/profiles/pmc_latest_sweep_kernel.json
cp $O/pmc_summary_sweep_group.json $R/profiles/pmc_latest_sweep_group_kernel.json
cd $R
GIPUMA_HIP_LAUNCH_TIMES=1 python bench.py > $O/bench_C.json 2> $O/bench_C.err
ls -la $O
