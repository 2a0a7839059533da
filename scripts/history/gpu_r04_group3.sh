export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_COUNTS=1 GIPUMA_HIP_LAUNCH_TIMES=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "gipuma_hip"
