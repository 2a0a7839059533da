#!/bin/bash
# round 5: box 19 (the reference's default window) -- parity of the new compile-time instantiation and its speed against
# the runtime-sized loop it replaces; views in flight on configs A and B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fast_mode.py -m gpu -x -q -k "full_run_bit_exact or other_boxes or production or fast_session or leak" > gpurun_out/r05_box19_pytest.txt 2>&1
tail -3 gpurun_out/r05_box19_pytest.txt
python bench.py --blocksize 19 --steps 3 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_box19.json 2> gpurun_out/r05_bench_box19.err
GIPUMA_HIP_EXPERIMENTS=1 GIPUMA_HIP_TUNE=8 python bench.py --blocksize 19 --steps 3 --no-cpu-baseline --no-extras > gpurun_out/r05_bench_box19_generic.json 2> gpurun_out/r05_bench_box19_generic.err
python bench.py --config B --steps 20 --no-cpu-baseline > gpurun_out/r05_bench_B.json 2> gpurun_out/r05_bench_B.err
python bench.py --config A --steps 50 --no-cpu-baseline > gpurun_out/r05_bench_A.json 2> gpurun_out/r05_bench_A.err
python - <<'PY'
import json
for n in ("box19", "box19_generic", "B", "A"):
    try:
        j = json.load(open("gpurun_out/r05_bench_%s.json" % n))
        print(n, "value %.3f ms %.3f" % (j["value"], j["ms_per_step"]), "fast", (j.get("value_fast") or {}).get("value"),
              (j.get("value_fast") or {}).get("parity_vs_exact_mode", {}).get("frac_within_tolerance"),
              "in flight", {k: round(v["value"], 2) for k, v in (j.get("value_views_in_flight") or {}).items() if isinstance(v, dict)},
              "exh", (j.get("value_exhaustive") or {}).get("value"), j["quality"])
    except Exception as e:
        print(n, "FAILED", e)
PY
