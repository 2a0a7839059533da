#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# One gpurun call that regenerates the round's evidence: kernel trace + per-launch series, PMC passes,
# bench lines for config C (default, with CPU baselines and extras), B, D and colour.
#   sh scripts/final_evidence.sh   ->  gpurun_out/final/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
sh $R/scripts/gpu_prof.sh final_prof GIPUMA_HIP_LAUNCH_TIMES=1 > $O/prof.txt 2>&1
cp $R/gpurun_out/final_prof/kernel_stats.txt $R/gpurun_out/final_prof/series_named.txt $R/gpurun_out/final_prof/err.txt $O/ 2>/dev/null
sh $R/scripts/pmc_passes.sh final_pmc > $O/pmc.txt 2>&1
cp $R/gpurun_out/final_pmc/pmc_summary*.json $O/ 2>/dev/null
# the bench line imports these counters: hand it the ones just collected (the same files are then committed)
cp $O/pmc_summary.json $R/profiles/pmc_latest.json
cp $O/pmc_summary_sweep_group.json $R/profiles/pmc_latest_sweep_group_kernel.json
cd $R
GIPUMA_HIP_LAUNCH_TIMES=1 python bench.py --steps 20 --warmup 2 > $O/bench_C.json 2> $O/bench_C.err
python bench.py --config B --steps 20 --no-cpu-baseline --no-extras > $O/bench_B.json 2> $O/bench_B.err
python bench.py --config D --steps 3 --no-cpu-baseline --no-extras > $O/bench_D.json 2> $O/bench_D.err
python bench.py --colour --steps 3 --no-cpu-baseline --no-extras > $O/bench_colour.json 2> $O/bench_colour.err
# kernel tables and per-launch series of config D and colour
BENCH_ARGS="--config D" sh $R/scripts/gpu_prof.sh final_prof_D > $O/prof_D.txt 2>&1
BENCH_ARGS="--colour" sh $R/scripts/gpu_prof.sh final_prof_colour > $O/prof_colour.txt 2>&1
ls -la $O
