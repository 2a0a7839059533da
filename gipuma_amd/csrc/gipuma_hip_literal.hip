// gipuma_hip_literal.hip -- the reference-order flavour of the library (GIPUMA_HIP_FLAG_LITERAL, include/gipuma_hip.h):
// gipuma_hip.hip compiled a third time with the per-sample arithmetic in the literal operation order of the reference's
// source (PM_LITERAL = 1, pm_core.h / pm_cost.h: view_cost_loop) into the same shared object.  Device code lives in namespace
// pm_lit, the entry points are gipuma_hipl_* with hidden visibility -- the exported C-ABI is the exact flavour's, which
// forwards the calls on a literal session here.
#include <hip/hip_runtime.h>

#define PM_LITERAL 1
#define GIPUMA_HIP_FLAVOUR_TU 1
#define pm pm_lit
#define gipuma_hip_session gipuma_hipl_session
#define gipuma_hip_version gipuma_hipl_version
#define gipuma_hip_last_error gipuma_hipl_last_error
#define gipuma_hip_device_count gipuma_hipl_device_count
#define gipuma_hip_cache_clear gipuma_hipl_cache_clear
#define gipuma_hip_selftest_reciprocal gipuma_hipl_selftest_reciprocal
#define gipuma_hip_selftest_quotient gipuma_hipl_selftest_quotient
#define gipuma_hip_create gipuma_hipl_create
#define gipuma_hip_destroy gipuma_hipl_destroy
#define gipuma_hip_init_planes gipuma_hipl_init_planes
#define gipuma_hip_sweep gipuma_hipl_sweep
#define gipuma_hip_finalize gipuma_hipl_finalize
#define gipuma_hip_eval_cost gipuma_hipl_eval_cost
#define gipuma_hip_get_state gipuma_hipl_get_state
#define gipuma_hip_set_state gipuma_hipl_set_state
#define gipuma_hip_state_device_ptrs gipuma_hipl_state_device_ptrs
#define gipuma_hip_solve gipuma_hipl_solve
#define gipuma_hip_launch_times gipuma_hipl_launch_times
#define gipuma_hip_group_times gipuma_hipl_group_times
#define gipuma_hip_schedule gipuma_hipl_schedule
#define gipuma_hip_run gipuma_hipl_run
#include "gipuma_hip.hip"
