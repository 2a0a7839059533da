#!/bin/sh
export GIPUMA_HIP_EXPERIMENTS=1  # the library reads its A/B switches only under this one
# Collect PMC passes for the PatchMatch kernels only, one rocprofv3 run per counter group
# (counters only with --kernel-trace; never together with sys/hip/hsa tracing), summarise on the
# box and keep just the small JSON (the rocpd databases are deleted: gpurun copies back <= 64 MiB).
#   sh scripts/pmc_passes.sh <name> [bench args...]   ->  gpurun_out/<name>/pmc_summary.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --kernel-include-regex "pm::" --pmc $group -d $OUT -o pmc$i -- \
     python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT/pmc$i.bench.json 2> $OUT/pmc$i.err
  echo "pass $i ($group): rc=$?"
done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
FETCH_SIZE
WRITE_SIZE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU
SQ_CYCLES SQ_THREAD_CYCLES_VALU
LIST
# per half-sweep of a view (16: a sweep kernel each + pm::push_kernel where the costs are pushed), then
# each kernel on its own (mean per dispatch)
python $R/scripts/rocprof_summary.py pmcunits $OUT ${PMC_UNITS:-16} sweep_ push_kernel group_kernel > $OUT/pmc_summary.json
python $R/scripts/rocprof_summary.py pmc $OUT push_kernel > $OUT/pmc_summary_push.json
python $R/scripts/rocprof_summary.py pmc $OUT "pm::group_kernel" > $OUT/pmc_summary_group.json
python $R/scripts/rocprof_summary.py pmc $OUT sweep_group_kernel > $OUT/pmc_summary_sweep_group.json
python $R/scripts/rocprof_summary.py pmc $OUT sweep_kernel > $OUT/pmc_summary_pixel_per_lane.json
python $R/scripts/rocprof_summary.py pmc $OUT sweep_cols_kernel > $OUT/pmc_summary_column_per_lane.json
python $R/scripts/rocprof_summary.py pmc $OUT init_kernel > $OUT/pmc_summary_init.json
# which binary and which source these counters belong to: bench.py flags imported figures as stale when the
# library it runs is not this one
python - $OUT $R <<'PY'
import glob, hashlib, json, os, sys
out, root = sys.argv[1:3]
lib = os.environ.get("GIPUMA_HIP_LIB") or os.path.join(root, "gipuma_amd", "csrc", "libgipuma_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
src = hashlib.sha256(b"".join(open(f, "rb").read() for f in sorted(glob.glob(os.path.join(root, "gipuma_amd", "csrc", "*.h*"))))).hexdigest()[:16]
commit = os.environ.get("GIPUMA_COMMIT", "")
for f in glob.glob(os.path.join(out, "pmc_summary*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    d["_lib_sha16"], d["_src_sha16"], d["_commit"] = sha, src, commit
    json.dump(d, open(f, "w"), indent=1, sort_keys=True)
PY
rm -f $OUT/*.db
ls -la $OUT | head -30
