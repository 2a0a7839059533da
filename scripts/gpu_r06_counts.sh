#!/bin/sh
# phase clocks and event counters of a config-C view (GIPUMA_HIP_COUNTS=1): where the fused plane-keyed launches spend their time
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_COUNTS=1
python - <<'PY'
from gipuma_amd import synth
from gipuma_amd.problem import Session
gs, _ = synth.build_problem("C")
with Session(gs) as s:
    s.solve(timing=True)
    t = s.solve(timing=True)
    print("ms", t.ms_total, [round(x, 2) for x in s.launch_times()[0]])
PY
