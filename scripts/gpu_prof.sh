#!/bin/sh
# rocprofv3 kernel trace of one bench step; prints the per-kernel table and the per-dispatch series
#   [BENCH_ARGS='--config D'] sh scripts/gpu_prof.sh <name> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $OUT/bench.json 2> $OUT/err.txt
DB=$(ls $OUT/*_results.db 2>/dev/null | head -1)
python $R/scripts/rocprof_summary.py stats $DB > $OUT/kernel_stats.txt
python $R/scripts/rocprof_summary.py series $DB "pm::" > $OUT/series_all.txt
python - $DB > $OUT/series_named.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, d in cur.execute("select name, duration from kernels where name like '%pm::%' order by start"):
    print("%-40s %8.3f" % (name.split("(")[0].replace("void pm::", "")[:40], d / 1e6))
PY
rm -f $OUT/*.db
cat $OUT/kernel_stats.txt
cat $OUT/series_named.txt | grep -v "pack_kernel\|check_u8"
