#!/bin/bash
# round 5: one rocprofv3 kernel trace of a bench run WITH its extra legs: the kernels of the three flavours side by side
# (pm:: exact, pm_fast:: GIPUMA_HIP_FLAG_FAST, pm_lit:: GIPUMA_HIP_FLAG_LITERAL)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_modes_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
DB=$(ls $OUT/*_results.db 2>/dev/null | head -1)
python - $DB > $OUT/kernel_stats_by_flavour.txt <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = collections.OrderedDict()
for name, d in cur.execute("select name, duration from kernels where name like '%pm%::%' order by start"):
    short = name.split("(")[0].replace("void ", "")
    r = rows.setdefault(short, [0, 0.0])
    r[0] += 1; r[1] += d / 1e6
print("%-60s %6s %10s %10s" % ("kernel (all launches of one bench run with its extra legs)", "calls", "total ms", "avg ms"))
for k, (n, t) in sorted(rows.items(), key=lambda kv: (kv[0].split("::")[0], -kv[1][1])):
    print("%-60s %6d %10.2f %10.3f" % (k[:60], n, t, t / n))
PY
rm -f $OUT/*.db
cat $OUT/kernel_stats_by_flavour.txt
