#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# A/B runs of bench.py on the GPU box: each line of stdin is "<label> <ENV=VAL ...>"; prints the
# bench JSON's value / ms and the last step's per-launch times.   sh scripts/gpu_ab.sh [bench args] < list
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab
while read -r label envs; do
  [ -z "$label" ] && continue
  env GIPUMA_HIP_LAUNCH_TIMES=1 $envs python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" \
      > $R/gpurun_out/ab/$label.json 2> $R/gpurun_out/ab/$label.err
  python - "$label" $R/gpurun_out/ab/$label.json $R/gpurun_out/ab/$label.err <<'PY'
import json, sys
lab, j, e = sys.argv[1:4]
try:
    d = json.load(open(j))
    lt = [l for l in open(e) if l.startswith("gipuma_hip launch_ms:")]
    print("%-14s %7.3f Mpix/s  %7.2f ms/step  dev %7.2f ms | %s" % (lab, d["value"], d["ms_per_step"],
          d["config"]["device_ms_total"], lt[-1].split(":", 1)[1].strip() if lt else ""))
except Exception as ex:
    print(lab, "FAILED", ex, open(e).read()[-400:])
PY
done
