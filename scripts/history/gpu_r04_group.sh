#!/bin/sh
# round 4: plane-keyed propagation (pm_group.h), wave-autonomous batches -- correctness + A/B on config C
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b; mkdir -p $O
V=$R/gipuma_amd/csrc/variants
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest_spl1.txt 2>&1; echo "pytest spl1 rc=$?"
GIPUMA_HIP_LIB=$V/libgipuma_hip_spl2.so timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest_spl2.txt 2>&1; echo "pytest spl2 rc=$?"
tail -3 $O/pytest_spl1.txt $O/pytest_spl2.txt
sh scripts/gpu_ab.sh <<LIST
base
g4_spl1 GIPUMA_HIP_GROUP_FROM=4
g4_spl2 GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_LIB=$V/libgipuma_hip_spl2.so
g6_spl1 GIPUMA_HIP_GROUP_FROM=6
g8_spl1 GIPUMA_HIP_GROUP_FROM=8
g4_spl1_counts GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_COUNTS=1
LIST
grep "phase ticks\|tasks/px" $R/gpurun_out/ab/g4_spl1_counts.err | head -5
# full-frame bit-equality with the exhaustive schedule, grouped propagation on
GIPUMA_HIP_GROUP_FROM=4 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_group_full.json 2> $O/bench_group_full.err
python - $O/bench_group_full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("group from 4: value %.3f  default_equals_exhaustive %s" % (d["value"], d.get("quality", {}).get("default_equals_exhaustive")))
PY
