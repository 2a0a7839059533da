#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d; mkdir -p $O
V=$R/gipuma_amd/csrc/variants
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k "plane_keyed" > $O/pytest.txt 2>&1; echo "pytest rc=$?"
sh scripts/gpu_ab.sh <<LIST
base
g4 GIPUMA_HIP_GROUP_FROM=4
g4_wg2 GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_LIB=$V/libgipuma_hip_wg2.so
g4_lanes2 GIPUMA_HIP_GROUP_FROM=4 GIPUMA_HIP_LIB=$V/libgipuma_hip_lanes2.so
base2
g4b GIPUMA_HIP_GROUP_FROM=4
LIST
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
LIST
python $R/scripts/rocprof_summary.py pmc $O group_kernel > $O/pmc_group_kernel.json
python $R/scripts/rocprof_summary.py stats $O/pmc1_results.db > $O/kernel_stats.txt 2>&1
rm -f $O/*.db
head -6 $O/kernel_stats.txt
