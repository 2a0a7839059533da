#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# push kernel on boxes 11 / 25: parity tests, then A/B timing on configs D, B and the generic loop on C
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/push2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "push" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
sh scripts/gpu_ab.sh > $O/abC.txt 2>&1 <<'AB'
c_def GIPUMA_HIP_PUSH_LAUNCHES=4
c_gen15 GIPUMA_HIP_LIB=gipuma_amd/csrc/variants/libgipuma_hip_gen15.so
AB
cat $O/abC.txt
sh scripts/gpu_ab.sh --config D --steps 2 > $O/abD.txt 2>&1 <<'AB'
d_p0 GIPUMA_HIP_PUSH_LAUNCHES=0
d_p3 GIPUMA_HIP_PUSH_LAUNCHES=3
d_p4 GIPUMA_HIP_PUSH_LAUNCHES=4
d_p6 GIPUMA_HIP_PUSH_LAUNCHES=6
d_p16 GIPUMA_HIP_PUSH_LAUNCHES=16
AB
cat $O/abD.txt
sh scripts/gpu_ab.sh --config B --steps 10 > $O/abB.txt 2>&1 <<'AB'
b_p0 GIPUMA_HIP_PUSH_LAUNCHES=0
b_p2 GIPUMA_HIP_PUSH_LAUNCHES=2
b_p4 GIPUMA_HIP_PUSH_LAUNCHES=4
b_p16 GIPUMA_HIP_PUSH_LAUNCHES=16
AB
cat $O/abB.txt
