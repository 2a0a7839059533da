#!/bin/bash
# round 5, first GPU call: the tolerance-judged mode against the exact one and against the reference's own code
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export GIPUMA_HIP_EXPERIMENTS=0
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1
timeout 600 python scripts/fast_mode_report.py --ref C:320x256 A:320x256 B:320x256 > gpurun_out/r05_fast_report_ref.txt 2>&1
timeout 300 python scripts/fast_mode_report.py A B C > gpurun_out/r05_fast_report_full.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "tiny or full_run or kernel_variants or plane_keyed" > gpurun_out/r05_pytest_subset.txt 2>&1
tail -3 gpurun_out/r05_pytest_subset.txt
cat gpurun_out/r05_fast_report_ref.txt gpurun_out/r05_fast_report_full.txt
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_bench_a.json"))
print("value", j["value"], "ms", j["ms_per_step"], "half", j["roofline"]["half_sweep_ms"])
print("fast", j.get("value_fast"))
print("exh", j["quality"])
PY
