#!/bin/bash
# round 5: the fused colour kernel (pm::sweep_group_kernel<15, 4>, not built since round 4: wrong costs "depending on code
# generation") rebuilt from the current sources under -DPM_FUSED_COLOUR_EXPERIMENT: is it still wrong, and does keeping scalar
# spills out of vector lanes (-mllvm -amdgpu-spill-sgpr-to-vgpr=0) change that?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/r05_fusedcolour.txt
for v in fusedc fusedc_nosv; do
  L=gipuma_amd/csrc/variants/libgipuma_hip_$v.so
  [ -f $L ] || continue
  echo "== $v: parity" >> gpurun_out/r05_fusedcolour.txt
  GIPUMA_HIP_LIB=$PWD/$L timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "plane_keyed_propagation_colour or colour_every_launch or (all_iterations and C-True)" >> gpurun_out/r05_fusedcolour.txt 2>&1
  echo "== $v: fused against two launches (the two-launch schedule equals the oracle in the test suite), whole frames" >> gpurun_out/r05_fusedcolour.txt
  GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_LIB=$PWD/$L timeout 300 python - >> gpurun_out/r05_fusedcolour.txt 2>&1 <<'PY'
import os, numpy as np
from gipuma_amd import synth
from gipuma_amd.problem import runcuda
for over in (dict(cols=832, rows=640), {}):
    gs, _ = synth.build_problem("C", colour=True, **over)
    out = {}
    for f in ("0", "1"):
        os.environ["GIPUMA_HIP_GROUP_FUSED"] = f
        out[f] = runcuda(gs)
    same_p = (out["0"][0].view(np.uint32) == out["1"][0].view(np.uint32)).all(-1).mean()
    same_c = (out["0"][1].view(np.uint32) == out["1"][1].view(np.uint32)).mean()
    print("colour %dx%d: planes identical %.6f, costs identical %.6f" % (gs.cols, gs.rows, same_p, same_c))
PY
  echo "== $v: colour bench, fused / two launches" >> gpurun_out/r05_fusedcolour.txt
  for f in 1 0; do
    GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_GROUP_FUSED=$f GIPUMA_HIP_LIB=$PWD/$L timeout 300 python bench.py --colour --steps 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('fused=$f', round(j['value'],3), 'Mpix/s', round(j['ms_per_step'],2), 'ms', j['schedule'], j['quality'])" >> gpurun_out/r05_fusedcolour.txt 2>&1
  done
done
cat gpurun_out/r05_fusedcolour.txt | grep -v "^$" | tail -40
