#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# colour push kernel: parity tests, then A/B timing on the colour variant of config C
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/push3
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "push_propagation_colour or colour_full_run" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
sh scripts/gpu_ab.sh --colour --steps 2 > $O/ab.txt 2>&1 <<'AB'
c_p0 GIPUMA_HIP_PUSH_LAUNCHES=0
c_p2 GIPUMA_HIP_PUSH_LAUNCHES=2
c_p4 GIPUMA_HIP_PUSH_LAUNCHES=4
c_p6 GIPUMA_HIP_PUSH_LAUNCHES=6
c_p8 GIPUMA_HIP_PUSH_LAUNCHES=8
c_p16 GIPUMA_HIP_PUSH_LAUNCHES=16
AB
cat $O/ab.txt
