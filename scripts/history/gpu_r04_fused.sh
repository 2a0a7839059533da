#!/bin/sh
# round 4: plane-keyed propagation fused with the sweep (pm::sweep_group_kernel) vs two launches vs none
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed or kernel_variants or fused_sweep or history_rule" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
sh scripts/gpu_ab.sh <<LIST
nogroup GIPUMA_HIP_GROUP_FROM=-1
unfused GIPUMA_HIP_GROUP_FUSED=0
fused
nogroup2 GIPUMA_HIP_GROUP_FROM=-1
unfused2 GIPUMA_HIP_GROUP_FUSED=0
fused2
LIST
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
python - $O/bench_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("fused default: value %.3f  default_equals_exhaustive %s steps %.2f patchy %.2f" % (d["value"], d.get("quality", {}).get("default_equals_exhaustive"), d.get("value_scene_steps",{}).get("value",0), d.get("value_scene_patchy",{}).get("value",0)))
PY
