#!/bin/sh
# round 4: plane-keyed propagation, second measurement: A/B + PMC counters of pm::group_kernel
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest.txt 2>&1; echo "pytest rc=$?"
sh scripts/gpu_ab.sh <<LIST
base
g4 GIPUMA_HIP_GROUP_FROM=4
g4_counts GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_COUNTS=1
LIST
grep "phase ticks" $R/gpurun_out/ab/g4_counts.err | tail -1
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i+1))
  GIPUMA_HIP_GROUP_FROM=4 timeout 240 rocprofv3 --kernel-trace --kernel-include-regex "pm::" --pmc $group -d $O -o pmc$i -- \
     python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc$i.bench.json 2> $O/pmc$i.err
  echo "pass $i ($group): rc=$?"
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
LIST
python $R/scripts/rocprof_summary.py pmc $O group_kernel > $O/pmc_group_kernel.json
python $R/scripts/rocprof_summary.py pmc $O sweep_kernel > $O/pmc_sweep_kernel_refine_only.json
python $R/scripts/rocprof_summary.py stats $O/pmc1_results.db > $O/kernel_stats.txt 2>&1
rm -f $O/*.db
cat $O/pmc_group_kernel.json
cat $O/kernel_stats.txt | head -12
