#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# like gpu_ab.sh, but each line of stdin is "<label> [bench args and ENV=VAL ...]" (words containing '=' and no
# leading '-' go to the environment, the rest to bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab
while read -r label rest; do
  [ -z "$label" ] && continue
  envs=""; args=""
  for w in $rest; do case "$w" in -*) args="$args $w";; *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  env GIPUMA_HIP_LAUNCH_TIMES=1 $envs python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $args "$@" \
      > $R/gpurun_out/ab/$label.json 2> $R/gpurun_out/ab/$label.err
  python - "$label" $R/gpurun_out/ab/$label.json $R/gpurun_out/ab/$label.err <<'PY'
import json, sys
lab, j, e = sys.argv[1:4]
try:
    d = json.load(open(j))
    lt = [l for l in open(e) if l.startswith("gipuma_hip launch_ms:")]
    print("%-14s %7.3f Mpix/s  %7.2f ms/step | %s" % (lab, d["value"], d["ms_per_step"], lt[-1].split(":", 1)[1].strip() if lt else ""))
except Exception as ex:
    print(lab, "FAILED", ex, open(e).read()[-400:])
PY
done
