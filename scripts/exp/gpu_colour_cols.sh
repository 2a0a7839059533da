#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# colour column-per-lane kernels: parity tests, then A/B timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ccols
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "colour and not every_launch" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
sh scripts/gpu_ab.sh --colour --steps 2 > $O/ab.txt 2>&1 <<'AB'
cc0 GIPUMA_HIP_COLS_LAUNCHES=0
cc2 GIPUMA_HIP_COLS_LAUNCHES=2
cc4 GIPUMA_HIP_COLS_LAUNCHES=4
cc6 GIPUMA_HIP_COLS_LAUNCHES=6
nocols GIPUMA_HIP_TUNE=134217728
AB
cat $O/ab.txt
