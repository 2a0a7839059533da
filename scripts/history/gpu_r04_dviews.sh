#!/bin/sh
# round 4, review item 6: is config D's column-per-lane kernel held back by the L2 working set ACROSS views?
# the same frame with 20, 5 and 1 source views: time of the first half-sweeps per view
export GIPUMA_HIP_EXPERIMENTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i; mkdir -p $O
for v in 20 5 1; do
  echo "views $v"
  sh scripts/gpu_ab.sh --config D --views $v <<LIST
D_v$v GIPUMA_HIP_PUSH_LAUNCHES=0 GIPUMA_HIP_GROUP_FROM=-1
LIST
done
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "sweep_cols_kernel" --pmc $grp -d $O -o pmcD$i -- \
     python $R/bench.py --config D --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/pmcD$i.json 2> $O/pmcD$i.err
  echo "pass ($grp) rc=$?"
done
python $R/scripts/rocprof_summary.py pmc $O sweep_cols_kernel > $O/pmc_D_cols.json
rm -f $O/*.db
cat $O/pmc_D_cols.json
