#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/r05_pushocc.txt
for v in default pushalias pushw3; do
  if [ $v = default ]; then L=gipuma_amd/csrc/libgipuma_hip.so; else L=gipuma_amd/csrc/variants/libgipuma_hip_$v.so; fi
  [ -f $L ] || continue
  echo "== $v" >> gpurun_out/r05_pushocc.txt
  GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_LIB=$PWD/$L timeout 300 python scripts/exp/push_occupancy.py >> gpurun_out/r05_pushocc.txt 2>&1
done
cat gpurun_out/r05_pushocc.txt
