// gipuma_hip_fast.hip -- the tolerance-judged flavour of the library (GIPUMA_HIP_FLAG_FAST, include/gipuma_hip.h):
// gipuma_hip.hip compiled a second time with the approx arithmetic of pm_core.h (PM_APPROX = 1) into the same shared
// object.  Device code lives in namespace pm_fast, the entry points are gipuma_hipf_* with hidden visibility -- the
// exported C-ABI is the exact flavour's, which forwards the calls on a fast session here.
#include <hip/hip_runtime.h>

#define PM_APPROX 1
#define GIPUMA_HIP_FLAVOUR_TU 1
#define pm pm_fast
#define gipuma_hip_session gipuma_hipf_session
#define gipuma_hip_version gipuma_hipf_version
#define gipuma_hip_last_error gipuma_hipf_last_error
#define gipuma_hip_device_count gipuma_hipf_device_count
#define gipuma_hip_cache_clear gipuma_hipf_cache_clear
#define gipuma_hip_selftest_reciprocal gipuma_hipf_selftest_reciprocal
#define gipuma_hip_selftest_quotient gipuma_hipf_selftest_quotient
#define gipuma_hip_create gipuma_hipf_create
#define gipuma_hip_destroy gipuma_hipf_destroy
#define gipuma_hip_init_planes gipuma_hipf_init_planes
#define gipuma_hip_sweep gipuma_hipf_sweep
#define gipuma_hip_finalize gipuma_hipf_finalize
#define gipuma_hip_eval_cost gipuma_hipf_eval_cost
#define gipuma_hip_get_state gipuma_hipf_get_state
#define gipuma_hip_set_state gipuma_hipf_set_state
#define gipuma_hip_state_device_ptrs gipuma_hipf_state_device_ptrs
#define gipuma_hip_solve gipuma_hipf_solve
#define gipuma_hip_launch_times gipuma_hipf_launch_times
#define gipuma_hip_group_times gipuma_hipf_group_times
#define gipuma_hip_schedule gipuma_hipf_schedule
#define gipuma_hip_run gipuma_hipf_run
#include "gipuma_hip.hip"
