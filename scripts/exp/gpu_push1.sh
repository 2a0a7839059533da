#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# first GPU check of the push kernel: parity tests, then A/B timing on config C
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/push1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "push or history_rule_survives or config_c_every" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
sh scripts/gpu_ab.sh > $O/ab.txt 2>&1 <<'AB'
push0 GIPUMA_HIP_PUSH_LAUNCHES=0
push4 GIPUMA_HIP_PUSH_LAUNCHES=4
push2 GIPUMA_HIP_PUSH_LAUNCHES=2
push8 GIPUMA_HIP_PUSH_LAUNCHES=8
push16 GIPUMA_HIP_PUSH_LAUNCHES=16
push16c8 GIPUMA_HIP_PUSH_LAUNCHES=16 GIPUMA_HIP_COLS_LAUNCHES=8
AB
cat $O/ab.txt
