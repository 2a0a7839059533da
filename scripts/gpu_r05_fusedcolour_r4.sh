#!/bin/bash
# round 5: the device code of commit 84d43b2 (round 4, where the fused colour kernel gave wrong costs) with that kernel
# re-enabled, built with and without -mllvm -amdgpu-spill-sgpr-to-vgpr=0: fused against two launches on whole frames
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/r05_fusedcolour_r4.txt
for v in r4fc r4fc_nosv r4fc_o1; do
  L=gipuma_amd/csrc/variants/libgipuma_hip_$v.so
  [ -f $L ] || continue
  echo "== $v" >> gpurun_out/r05_fusedcolour_r4.txt
  GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_LIB=$PWD/$L timeout 300 python - >> gpurun_out/r05_fusedcolour_r4.txt 2>&1 <<'PY'
import os, numpy as np
from gipuma_amd import synth
from gipuma_amd.problem import runcuda
for over in (dict(cols=832, rows=640), {}):
    gs, _ = synth.build_problem("C", colour=True, **over)
    out = {}
    for f in ("0", "1"):
        os.environ["GIPUMA_HIP_GROUP_FUSED"] = f
        out[f] = runcuda(gs)
    same_p = (out["0"][0].view(np.uint32) == out["1"][0].view(np.uint32)).all(-1)
    same_c = (out["0"][1].view(np.uint32) == out["1"][1].view(np.uint32))
    print("colour %dx%d: planes identical %.6f, costs identical %.6f" % (gs.cols, gs.rows, same_p.mean(), same_c.mean()))
    if not same_p.all():
        ys, xs = np.nonzero(~same_p)
        print("   first differing pixels (y, x):", list(zip(ys[:6].tolist(), xs[:6].tolist())), " rows with differences: %d of %d" % (len(set(ys.tolist())), gs.rows))
PY
done
cat gpurun_out/r05_fusedcolour_r4.txt | grep -v "^$\|amdgpu.ids"
