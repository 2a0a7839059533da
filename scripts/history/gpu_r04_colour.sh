#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "seen_rule or colour" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
sh scripts/gpu_ab.sh --colour <<LIST
colour_seen
colour_noseen GIPUMA_HIP_TUNE=4194304
colour_seen2
LIST
