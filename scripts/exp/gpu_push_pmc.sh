#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# PMC passes over the push kernel's dispatches of one config-C view (every half-sweep pushed), per dispatch
#   sh scripts/gpu_push_pmc.sh <name> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i+1))
  env GIPUMA_HIP_PUSH_LAUNCHES=16 "$@" timeout 240 rocprofv3 --kernel-trace --kernel-include-regex "push_kernel" --pmc $group -d $OUT -o pmc$i -- \
     python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc$i.bench.json 2> $OUT/pmc$i.err
  echo "pass $i ($group): rc=$?"
  DB=$(ls $OUT/pmc$i*_results.db 2>/dev/null | head -1)
  python $R/scripts/rocprof_summary.py pmcseries $DB push_kernel > $OUT/series$i.txt
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
LIST
python $R/scripts/rocprof_summary.py pmc $OUT push_kernel > $OUT/pmc_summary_push.json
rm -f $OUT/*.db
cat $OUT/series*.txt
