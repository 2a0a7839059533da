#!/bin/bash
# round 5: GIPUMA_HIP_FLAG_LITERAL -- the reference-order flavour against the reference's own code, and what it costs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_literal_mode.py -m gpu -x -q --durations=8 > gpurun_out/r05_literal_pytest.txt 2>&1
tail -15 gpurun_out/r05_literal_pytest.txt
timeout 600 python - > gpurun_out/r05_literal_timing.txt 2>&1 <<'PY'
import numpy as np
from gipuma_amd import synth
from gipuma_amd.problem import Session
for cfg, over in (("B", {}), ("C", dict(cols=320, rows=256)), ("C", {})):
    gs, info = synth.build_problem(cfg, **over)
    out = {}
    for name, kw in (("exact", {}), ("literal", dict(literal=True))):
        with Session(gs, **kw) as s:
            s.solve(timing=True)
            t = s.solve(timing=True)
            out[name] = (s.get_state()[0], t.ms_total)
    d = np.abs(out["exact"][0][..., 3] - out["literal"][0][..., 3]) / np.maximum(np.abs(out["exact"][0][..., 3]), 1e-30)
    n = np.abs(out["exact"][0][..., :3] - out["literal"][0][..., :3]).max(-1)
    print("config %s %dx%d: exact %.2f ms, literal %.2f ms (%.1fx); literal vs exact mode: %.4f of the pixels inside 1e-4 / 1e-3"
          % (cfg, gs.cols, gs.rows, out["exact"][1], out["literal"][1], out["literal"][1] / out["exact"][1], ((d < 1e-4) & (n < 1e-3)).mean()), flush=True)
PY
cat gpurun_out/r05_literal_timing.txt | grep -v amdgpu.ids
