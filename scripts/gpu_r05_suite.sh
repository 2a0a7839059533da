#!/bin/bash
# round 5: the whole -m gpu suite + a bench line (+ optional extra commands given as arguments)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=${SUITE_TIMEOUT:-1100}
timeout $T python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_pytest_gpu.txt 2>&1
tail -25 gpurun_out/r05_pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_bench_b.json"))
print("value", j["value"], "ms", j["ms_per_step"], "half", [round(x, 2) for x in j["roofline"]["half_sweep_ms"]], "init", j["config"]["device_ms_init"])
f = j.get("value_fast", {})
print("fast", f.get("value"), f.get("ms_per_step"), f.get("parity_vs_exact_mode"))
print("quality", j["quality"], "exh", j.get("value_exhaustive", {}).get("value"), "steps", j.get("value_scene_steps", {}).get("value"), "patchy", j.get("value_scene_patchy", {}).get("value"))
PY
