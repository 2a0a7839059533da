#!/bin/bash
# round 5: the three modes against the reference's OWN code at the headline frame size (config C, 1600x1216; the dump comes
# from scripts/make_ref_headline_dump.py, 23 minutes of CPU, and travels with the snapshot)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python - > gpurun_out/r05_modes_vs_reference_headline.txt 2>&1 <<'PY'
import numpy as np
from gipuma_amd import synth
from gipuma_amd.problem import Session
ref = np.load("scratch_big/ref_configC_1600x1216.npz")
rn, rc = ref["norm4"], ref["cost"]
gs, info = synth.build_problem("C", cols=1600, rows=1216)
print("config C at 1600x1216 (10 source views, box 15, best-3, 8 iterations): the reference's own device code (CPU, fp32 filter weights) against the three modes on an MI355X")
print("%-8s %10s %14s %14s %14s" % ("mode", "ms / view", "in tolerance", "planes ==", "costs =="))
for name, kw in (("literal", dict(literal=True)), ("exact", {}), ("fast", dict(fast=True))):
    with Session(gs, **kw) as s:
        s.solve(timing=True)
        t = s.solve(timing=True)
        n4, c = s.get_state()
    d = np.abs(n4[..., 3] - rn[..., 3]) / np.maximum(np.abs(rn[..., 3]), 1e-30)
    n = np.abs(n4[..., :3] - rn[..., :3]).max(-1)
    same = (n4.view(np.uint32) == rn.view(np.uint32)).all(-1)
    csame = c.view(np.uint32) == rc.view(np.uint32)
    print("%-8s %10.2f %13.4f%% %13.4f%% %13.4f%%" % (name, t.ms_total, 100 * ((d < 1e-4) & (n < 1e-3)).mean(), 100 * same.mean(), 100 * csame.mean()), flush=True)
PY
grep -v amdgpu.ids gpurun_out/r05_modes_vs_reference_headline.txt
