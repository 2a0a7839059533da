#!/bin/sh
# The bounds-checked TEST build (-DPM_CHECKED, pm_core.h; GPU AddressSanitizer is not available on this pool) through the whole
# GPU suite and one solve of every bench configuration in every mode: every computed global index / window offset compared
# with the extent of its buffer.  One line per session in gpurun_out/r06_checked_sessions.log; the summary at the end.
#   sh scripts/build_variant.sh checked -DPM_CHECKED   (here)      sh scripts/gpu_r06_checked.sh   (GPU box)
cd "$(dirname "$0")/.." || exit 1
export GIPUMA_HIP_EXPERIMENTS=1
export GIPUMA_HIP_LIB=$PWD/gipuma_amd/csrc/variants/libgipuma_hip_checked.so
export GIPUMA_CHECKED_LOG=$PWD/gpurun_out/r06_checked_sessions.log
rm -f $GIPUMA_CHECKED_LOG
# (GIPUMA_CHECKED_LOG, not GIPUMA_HIP_*: tests/conftest.py scrubs that prefix)
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r06_checked_pytest.txt 2>&1
tail -14 gpurun_out/r06_checked_pytest.txt
python scripts/gpu_r06_time.py C D colour box19 box11 B A C@fast C@literal colour@literal D@literal 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import os, re
lines = [l for l in open(os.environ["GIPUMA_CHECKED_LOG"]) if "CHECKED session" in l]
tot = sum(int(re.search(r"violations (\d+)", l).group(1)) for l in lines)
kinds = {}
for l in lines:
    k = re.search(r"session (\S+ ch \d+ box \d+ views \d+)", l).group(1)
    kinds[k] = kinds.get(k, 0) + 1
print("checked build: %d sessions (%d distinct shapes), %d out-of-bounds accesses in total" % (len(lines), len(kinds), tot))
for l in lines:
    if not re.search(r"violations 0 ", l):
        print("  ", l.strip())
PY
