#!/bin/bash
# round 5: which ingredient of the tolerance-judged mode costs how much agreement / buys how much time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in default relay relaytaps; do
  if [ $v = default ]; then L=gipuma_amd/csrc/libgipuma_hip.so; else L=gipuma_amd/csrc/variants/libgipuma_hip_$v.so; fi
  [ -f $L ] || continue
  echo "== $v" >> gpurun_out/r05_fast_variants.txt
  GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_LIB=$PWD/$L timeout 300 python scripts/fast_mode_report.py C:320x256 B C >> gpurun_out/r05_fast_variants.txt 2>&1
done
cat gpurun_out/r05_fast_variants.txt
