#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# per-launch PMC series of one kernel: sh scripts/exp/gpu_pmc_series.sh <name> <kernel-substring> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; SUB=$2; shift; shift
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i+1))
  env "$@" timeout 240 rocprofv3 --kernel-trace --kernel-include-regex "pm::" --pmc $group -d $OUT -o pmc$i -- \
     python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc$i.bench.json 2> $OUT/pmc$i.err
  echo "pass $i ($group): rc=$?"
  db=$(ls $OUT/pmc${i}_results.db 2>/dev/null | head -1)
  python $R/scripts/rocprof_summary.py pmcseries $db $SUB > $OUT/series$i.txt
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
LIST
rm -f $OUT/*.db
cat $OUT/series*.txt
