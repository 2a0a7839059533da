// gipuma_hip.hip -- C-ABI (include/gipuma_hip.h) over the gfx950 kernels in pm_device.h.
//
// Host side of what the reference does in gipuma<T>() (gipuma.cu:1825-1960): validate, make the
// problem resident in HBM, launch init / red-black sweeps / finalize on one HIP stream, time with
// HIP events.  Differences from the reference that are deliberate:
//   * one launch per colour (close+far+refine fused, result-identical, see sweep_kernel) and no
//     host synchronisation between launches (the reference calls cudaDeviceSynchronize after
//     each of its 6 launches per iteration, gipuma.cu:1916-1936);
//   * cameras are packed once into one POD block read through scalar loads, instead of the
//     ~3600 managed allocations the reference dereferences on the device (camera.h:45-51);
//   * no per-pixel RNG state array (48 B/pixel, gipuma.cu:1840): the RNG is counter based.  Per-pixel state of a
//     session: 20 B planes + costs, 1 B history flag, and -- performance only, optional (the solve runs without
//     them when the allocation fails) -- 32 B pushed propagation costs and 16-64 B prefilter sample lists.
//
// This file is compiled THREE times into libgipuma_hip.so.  As itself it is the exact flavour (bit-identical to the CPU
// restatement of the numerical model) and owns the exported C-ABI.  Included by gipuma_hip_fast.hip (PM_APPROX = 1,
// namespace pm -> pm_fast, entry points gipuma_hipf_*) it is the tolerance-judged flavour behind GIPUMA_HIP_FLAG_FAST; included
// by gipuma_hip_literal.hip (PM_LITERAL = 1, pm_lit, gipuma_hipl_*) the reference-order flavour behind GIPUMA_HIP_FLAG_LITERAL.
// Same host logic in all three (GIPUMA_HIP_FLAVOUR_TU marks the two inclusions; their symbols have hidden visibility).  A
// session created with one of the flags is a thin handle whose calls this flavour forwards through a table of the other
// flavour's entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <tuple>
#include <string>
#include <type_traits>
#include <vector>

#ifdef GIPUMA_HIP_FLAVOUR_TU
#pragma GCC visibility push(hidden)
#endif
#include "../../include/gipuma_hip.h"
#include "pm_device.h"
#include "pm_push.h"
#include "pm_group.h"

#ifndef GIPUMA_HIP_FLAVOUR_TU
// the session entry points of the other flavours (C linkage: the session pointer is opaque here)
extern "C" {
#pragma GCC visibility push(hidden)
#define DECLARE_FLAVOUR(P)                                                                                             \
    const char *P##last_error(void);                                                                                    \
    int P##cache_clear(void);                                                                                           \
    int P##create(const gipuma_hip_desc *desc, void **out);                                                             \
    int P##destroy(void *s);                                                                                            \
    int P##init_planes(void *s);                                                                                        \
    int P##sweep(void *s, int iteration, int colour, unsigned stages);                                                  \
    int P##finalize(void *s);                                                                                           \
    int P##eval_cost(void *s, const float *planes_host, float *cost_out_host);                                          \
    int P##get_state(void *s, float *norm4_host, float *cost_host);                                                     \
    int P##set_state(void *s, const float *norm4_host, const float *cost_host);                                         \
    int P##state_device_ptrs(void *s, float **norm4_dev, float **cost_dev);                                             \
    int P##solve(void *s, gipuma_hip_timing *timing);                                                                   \
    int P##launch_times(void *s, float *ms_half_sweep, int capacity, int *n_half_sweeps, int *n_pushed);                \
    int P##group_times(void *s, float *ms_group, int capacity, int *n_half_sweeps);                                     \
    int P##schedule(void *s, int info[4]);
DECLARE_FLAVOUR(gipuma_hipf_)
DECLARE_FLAVOUR(gipuma_hipl_)
#undef DECLARE_FLAVOUR
#pragma GCC visibility pop
}
struct FlavourApi {
    const char *(*last_error)(void);
    int (*cache_clear)(void);
    int (*create)(const gipuma_hip_desc *, void **);
    int (*destroy)(void *);
    int (*init_planes)(void *);
    int (*sweep)(void *, int, int, unsigned);
    int (*finalize)(void *);
    int (*eval_cost)(void *, const float *, float *);
    int (*get_state)(void *, float *, float *);
    int (*set_state)(void *, const float *, const float *);
    int (*state_device_ptrs)(void *, float **, float **);
    int (*solve)(void *, gipuma_hip_timing *);
    int (*launch_times)(void *, float *, int, int *, int *);
    int (*group_times)(void *, float *, int, int *);
    int (*schedule)(void *, int *);
};
#define FLAVOUR_API(P)                                                                                                  \
    {P##last_error, P##cache_clear, P##create, P##destroy, P##init_planes, P##sweep, P##finalize, P##eval_cost,        \
     P##get_state, P##set_state, P##state_device_ptrs, P##solve, P##launch_times, P##group_times, P##schedule}
static const FlavourApi kFastApi = FLAVOUR_API(gipuma_hipf_), kLiteralApi = FLAVOUR_API(gipuma_hipl_);
#undef FLAVOUR_API
#else
struct FlavourApi;
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, const char *a = "", const char *b = "")
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    g_err = buf;
    return code;
}

// GIPUMA_HIP_FLAG_CACHE_IMAGES: what has been derived from a resident image plane, per
// (device, address, rows, cols, pitch, channels)
struct CachedImage {
    int not_u8 = -1;              // result of the 8-bit check (-1: not run yet)
    uint32_t *packed = nullptr;   // window-packed copy (pack_kernel / pack_kernel_c4)
    int users = 0;                // live sessions whose Problem points at `packed`
};
typedef std::tuple<int, const void *, int, int, int, int> CacheKey;
std::map<CacheKey, CachedImage> g_cache;
std::mutex g_cache_mutex;

// Experiment switches (A/B runs, tests of the work-reduction rules): every GIPUMA_HIP_<name> variable below is read
// ONLY when GIPUMA_HIP_EXPERIMENTS is set to a non-zero value -- a production process never changes its schedule on
// ambient environment variables.  None of them changes a result (tests/test_parity_gpu.py).
const char *exp_env(const char *name)
{
    const char *on = getenv("GIPUMA_HIP_EXPERIMENTS");
    if (!on || atoi(on) == 0) return nullptr;
    char buf[64];
    snprintf(buf, sizeof buf, "GIPUMA_HIP_%s", name);
    return getenv(buf);
}

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(GIPUMA_HIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#ifndef GIPUMA_HIP_FLAVOUR_TU
// forward a call on a GIPUMA_HIP_FLAG_FAST / _LITERAL session to its flavour; a failure's text becomes this thread's last error
#define FORWARD(s, fn, ...)                                     \
    do {                                                        \
        if ((s) && (s)->api) {                                  \
            const int rc_ = (s)->api->fn((s)->impl, ##__VA_ARGS__); \
            if (rc_) g_err = (s)->api->last_error();            \
            return rc_;                                         \
        }                                                       \
    } while (0)
#else
#define FORWARD(s, fn, ...) do { } while (0)
#endif

}  // namespace

struct gipuma_hip_session {
    // non-null: this object is only the handle of a GIPUMA_HIP_FLAG_FAST / _LITERAL session that lives in another flavour
    // of this file; every entry point forwards to it through `api` and nothing below is used
    const FlavourApi *api = nullptr;
    void *impl = nullptr;
    int device = 0;
    int rows = 0, cols = 0, n_sel = 0, iterations = 0;
    pm::Problem hp{};
    pm::Problem *dp = nullptr;
    float4 *norm4 = nullptr;
    float *cost = nullptr;
    std::vector<float *> owned;  // device copies of host images
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool u8 = false;         // every image integer valued in [0,255] -> weight table + packed windows
    std::vector<uint32_t *> packed;  // window-packed copies of the selected views (U8 mode)
    std::vector<CacheKey> cache_refs;  // ... or shared ones it holds a use count on (GIPUMA_HIP_FLAG_CACHE_IMAGES)
    int *flag = nullptr;
    bool combine_reg = false;
    bool unfused = false;
    // state invariant cost[p] == cost(p, plane[p]): true once init_planes has run, not assumed after
    // gipuma_hip_set_state (the caller may install any pair); the sweep kernel's skip rule (A) needs it
    bool costs_trusted = false;
    // gipuma_hip_finalize rewrites norm4 in place to (world normal, depth): no sweep may follow until the
    // planes are re-initialised or re-installed
    bool finalized = false;
    // history rule bookkeeping: colours of the last two launches that were full-stage, fused, trusted
    // half-sweeps (-1 otherwise); the rule is valid for colour c iff prev1 == 1-c and prev2 == c
    int prev1 = -1, prev2 = -1;
    unsigned char *changed = nullptr;  // device, one byte per pixel
    unsigned *et_stat = nullptr;       // device, Problem::et_stat
    unsigned char *et_hint = nullptr;  // device, 12 bytes per sweep tile (Problem::et_hint)
    // lower-bound prefilter of refinement candidates (pm::lb_item): the heaviest window samples of every
    // pixel, listed by pm::weight_order_kernel at the start of every solve (init_planes) or before the
    // first sweep that needs them
    unsigned long long *dbg = nullptr;  // device, Problem::dbg (GIPUMA_HIP_COUNTS=1)
    unsigned long long *tile_clock = nullptr;  // device, Problem::tile_clock (dispatch order of the fused launches)
    int *tile_order = nullptr;                 // device, Problem::tile_order
#ifdef PM_WG_TICKS
    unsigned long long *wg_ticks = nullptr;  // (experiment build) GIPUMA_HIP_WG_TICKS=<file>: per-workgroup clocks of the fused launches
#endif
#ifdef PM_CHECKED
    unsigned long long *viol = nullptr;  // device, Problem::viol
#endif
    float4 *seen_ring = nullptr;        // device, Problem::seen_ring (skip rule (S), colour sessions)
    unsigned char *seen_pos = nullptr;  // device, Problem::seen_pos
    uint32_t *worder = nullptr;  // device, Problem::worder
    bool worder_valid = false;
    size_t et_hint_bytes = 0;
    // push propagation (pm_push.h): after a half-sweep the planes of its colour are evaluated once for
    // all their consumers; the next half-sweep reads those costs instead of evaluating them
    bool push_ok = false;      // the instantiation exists for this problem
    int push_launches = 0;     // leading half-sweeps (2*iteration + colour) that consume pushed costs
    float *push_cost = nullptr;  // device, Problem::push_cost
    int push_valid = -1;       // colour whose pixels find valid costs in push_cost (-1: nobody)
    bool push_hist = false;    // ... offered under rule (H) (only the planes that changed)
    bool push_attr_set = false;
    // plane-keyed propagation (pm_group.h): from half-sweep `group_from` on the propagation costs of a half-sweep
    // come from pm::group_kernel, launched right before it
    bool group_ok = false;
    int group_from = 0;
    bool group_attr_set = false;
    bool group_fused = true;   // one launch per half-sweep (pm::sweep_group_kernel) instead of group_kernel + sweep_kernel
    bool fused_attr_set = false;
    int box = 0;             // specialised window size, 0 = runtime
    int ch = 1;              // 1 = gray (T=float), 4 = colour (T=float4)
    unsigned tune = 0;
    int cols_launches = -1;  // leading half-sweeps (2*iteration + colour) evaluated column-per-lane (-1: by box)
    size_t lds_sweep = 0, lds_dense = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // experiment aid (GIPUMA_HIP_LAUNCH_TIMES=1): one event per half-sweep launch of gipuma_hip_solve,
    // durations printed to stderr
    bool launch_times = false;
    std::vector<hipEvent_t> lev;
    std::vector<float> half_sweep_ms;  // of the last timed gipuma_hip_solve (gipuma_hip_launch_times)
    std::vector<hipEvent_t> gev;       // [2 * half-sweep]: around the pm::group_kernel launch of a half-sweep
    std::vector<char> gev_used;        // per half-sweep: it had such a launch
    std::vector<float> group_ms;       // of the last timed solve (gipuma_hip_group_times)
    int timed_half_sweep = -1;         // >= 0 while a timed solve is launching that half-sweep
    int n_pushed = 0;                  // leading half-sweeps of that solve that read pushed costs
    int n_push_consumed = 0;           // ... counted while the solve runs
};

namespace {

using pm::Tune;

typedef void (*sweep_fn)(const pm::Problem *, float4 *, float *, int, uint32_t, unsigned, unsigned);
typedef void (*init_fn)(const pm::Problem *, float4 *, float *, unsigned);

template <int BOX, int CH>
sweep_fn pick_sweep_box(bool u8, bool creg)
{
    if (u8) return creg ? pm::sweep_kernel<BOX, true, true, true, CH> : pm::sweep_kernel<BOX, true, false, true, CH>;
    return creg ? pm::sweep_kernel<BOX, false, true, true, CH> : pm::sweep_kernel<BOX, false, false, true, CH>;
}

template <int CH>
sweep_fn pick_sweep_ch(const gipuma_hip_session *s)
{
    switch (s->box) {
    case 11: return pick_sweep_box<11, CH>(s->u8, s->combine_reg);
    case 15: return pick_sweep_box<15, CH>(s->u8, s->combine_reg);
    case 19:  // the reference's default window (algorithmparameters.h:25-26), gray: since round 6 with every kernel family of
              // boxes 15 and 25 (push, column-per-lane in groups of 16 lanes, plane-keyed, prefilter)
        if constexpr (CH == 1) return pick_sweep_box<19, 1>(s->u8, s->combine_reg);
        return pick_sweep_box<0, CH>(s->u8, s->combine_reg);
    case 25: return pick_sweep_box<25, CH>(s->u8, s->combine_reg);
    default: return pick_sweep_box<0, CH>(s->u8, s->combine_reg);
    }
}

sweep_fn pick_sweep(const gipuma_hip_session *s)
{
    if (s->ch == 4) return pick_sweep_ch<4>(s);
    if (s->tune & Tune::kNoInterior) {  // A/B switch, gray only
        if (s->box == 15 && s->u8 && s->combine_reg) return pm::sweep_kernel<15, true, true, false, 1>;
        return s->u8 ? pm::sweep_kernel<0, true, false, false, 1> : pm::sweep_kernel<0, false, false, false, 1>;
    }
    return pick_sweep_ch<1>(s);
}

template <bool GEN, int CH>
init_fn pick_init_ch(const gipuma_hip_session *s)
{
    switch (s->box) {
    case 11: return s->u8 ? pm::init_kernel<11, true, false, GEN, CH> : pm::init_kernel<11, false, false, GEN, CH>;
    case 15: return s->u8 ? pm::init_kernel<15, true, false, GEN, CH> : pm::init_kernel<15, false, false, GEN, CH>;
    case 19:
        if constexpr (CH == 1) return s->u8 ? pm::init_kernel<19, true, false, GEN, 1> : pm::init_kernel<19, false, false, GEN, 1>;
        return s->u8 ? pm::init_kernel<0, true, false, GEN, CH> : pm::init_kernel<0, false, false, GEN, CH>;
    case 25: return s->u8 ? pm::init_kernel<25, true, false, GEN, CH> : pm::init_kernel<25, false, false, GEN, CH>;
    default: return s->u8 ? pm::init_kernel<0, true, false, GEN, CH> : pm::init_kernel<0, false, false, GEN, CH>;
    }
}

template <bool GEN>
init_fn pick_init(const gipuma_hip_session *s)
{
    return s->ch == 4 ? pick_init_ch<GEN, 4>(s) : pick_init_ch<GEN, 1>(s);
}

int validate(const gipuma_hip_desc *d)
{
    if (!d) return fail(GIPUMA_HIP_ERR_ARG, "null descriptor");
    if (d->abi_version != GIPUMA_HIP_ABI_VERSION) return fail(GIPUMA_HIP_ERR_ARG, "abi_version mismatch");
    if (d->rows < 1 || d->cols < 1) return fail(GIPUMA_HIP_ERR_ARG, "rows/cols must be positive");
    if ((long long)d->rows * (long long)d->pitch >= (1LL << 29))
        return fail(GIPUMA_HIP_ERR_ARG, "image too large for 32-bit texel offsets");
    if (d->channels != 1 && d->channels != 4)
        return fail(GIPUMA_HIP_ERR_UNSUPPORTED, "channels must be 1 (gray, T=float) or 4 (colour, T=float4)");
    if (d->pitch < d->cols * d->channels) return fail(GIPUMA_HIP_ERR_ARG, "pitch < cols*channels");
    if (d->channels == 4 && (d->pitch & 3)) return fail(GIPUMA_HIP_ERR_ARG, "colour pitch must be a multiple of 4 floats");
    if (d->n_images < 1 || d->n_images > 512 || !d->images || !d->cameras)
        return fail(GIPUMA_HIP_ERR_ARG, "images/cameras missing");
    if (d->n_selected < 0 || d->n_selected > GIPUMA_HIP_MAX_VIEWS || (d->n_selected > 0 && !d->selected))
        return fail(GIPUMA_HIP_ERR_ARG, "n_selected must be 0..32 (gipuma.cu:736)");
    for (int i = 0; i < d->n_selected; i++)
        if (d->selected[i] < 0 || d->selected[i] >= d->n_images || !d->images[d->selected[i]])
            return fail(GIPUMA_HIP_ERR_ARG, "selected view out of range");
    if (!d->images[0]) return fail(GIPUMA_HIP_ERR_ARG, "reference image missing");
    const gipuma_hip_params &p = d->params;
    if (p.box_hsize < 1 || p.box_vsize < 1 || !(p.box_hsize & 1) || !(p.box_vsize & 1))
        return fail(GIPUMA_HIP_ERR_ARG, "box sizes must be odd (main.cpp:269-276)");
    if (p.box_hsize > 49 || p.box_vsize > 49) return fail(GIPUMA_HIP_ERR_UNSUPPORTED, "box size > 49");
    if (p.iterations < 0) return fail(GIPUMA_HIP_ERR_ARG, "iterations < 0");
    return 0;
}

void copy9(float *dst, const float *src) { memcpy(dst, src, 9 * sizeof(float)); }
void copy3(float *dst, const float *src) { memcpy(dst, src, 3 * sizeof(float)); }

size_t lds_bytes(const gipuma_hip_session *s, int tile_h, bool with_cv, bool sweep)
{
    const int hw = (s->hp.box_h + 1) / 2, hh = (s->hp.box_v + 1) / 2;
    const int texels = (pm::kTileW + 2 * hw) * (tile_h + 2 * hh);
    size_t n = (s->ch == 4 ? pm::lut_size<4>() : pm::lut_size<1>()) + (size_t)4 * texels +
               (size_t)(s->ch == 4 ? pm::work_floats<4>(texels, sweep) : pm::work_floats<1>(texels, sweep));
    if (with_cv) n += (size_t)s->n_sel * pm::kThreads;
    return n * sizeof(float);
}

// pm::push_kernel: the planes of `colour` evaluated for their consumers (the pixels of the other colour)
int launch_push(gipuma_hip_session *s, int colour, bool hist)
{
    const int gx = (s->cols + pm::kTileW - 1) / pm::kTileW;
    const int gy = (s->rows + pm::kSweepTileH - 1) / pm::kSweepTileH;
    typedef void (*push_fn)(const pm::Problem *, const float4 *, int, int, unsigned);
    const push_fn k = s->ch == 4    ? pm::push_kernel_c4<15>
                      : s->box == 15 ? pm::push_kernel<15>
                      : s->box == 19 ? pm::push_kernel<19>
                      : s->box == 25 ? pm::push_kernel<25>
                                     : pm::push_kernel<11>;
    size_t lds = sizeof(float) * (size_t)(s->ch == 4      ? pm::PushLayoutC4<15>::total
                                          : s->box == 15 ? pm::PushLayout<15>::total
                                          : s->box == 19 ? pm::PushLayout<19>::total
                                          : s->box == 25 ? pm::PushLayout<25>::total
                                                         : pm::PushLayout<11>::total);
    if (const char *t = exp_env("PUSH_LDS_KB")) lds = std::max(lds, (size_t)atoi(t) * 1024);  // experiment: fewer workgroups per CU
    if (!s->push_attr_set) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        s->push_attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(gx * gy), dim3(pm::kThreads), lds, s->stream, s->dp, s->norm4, colour, hist ? 1 : 0,
                       s->tune);
    HIP_OK(hipGetLastError());
    s->push_valid = 1 - colour;
    s->push_hist = hist;
    return 0;
}

// pm::group_kernel: the propagation costs of the half-sweep of `colour` that follows, one evaluation per plane
int launch_group(gipuma_hip_session *s, int colour, bool hist, unsigned tune)
{
    const int gx = (s->cols + pm::kTileW - 1) / pm::kTileW;
    const int gy = (s->rows + pm::kSweepTileH - 1) / pm::kSweepTileH;
    typedef void (*group_fn)(const pm::Problem *, const float4 *, const float *, int, int, unsigned);
    const group_fn k = s->ch == 4     ? pm::group_kernel<15, 4>
                       : s->box == 15 ? pm::group_kernel<15>
                       : s->box == 19 ? pm::group_kernel<19>
                       : s->box == 25 ? pm::group_kernel<25>
                                      : pm::group_kernel<11>;
    const size_t lds = sizeof(float) * (size_t)(s->ch == 4     ? pm::GroupLayout<15, 4>::total
                                                : s->box == 15 ? pm::GroupLayout<15>::total
                                                : s->box == 19 ? pm::GroupLayout<19>::total
                                                : s->box == 25 ? pm::GroupLayout<25>::total
                                                               : pm::GroupLayout<11>::total);
    if (!s->group_attr_set) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        s->group_attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(gx * gy), dim3(pm::kThreads), lds, s->stream, s->dp, s->norm4, s->cost, colour,
                       hist ? 1 : 0, tune & ~(Tune::kPushConsume | Tune::kHistorySkip));
    HIP_OK(hipGetLastError());
    s->push_valid = colour;
    s->push_hist = hist;
    return 0;
}

#ifndef PM_TILE_ORDER_DEFAULT
#define PM_TILE_ORDER_DEFAULT 0  // (the fused launches' dispatch order from the previous durations: off unless GIPUMA_HIP_TILE_ORDER=1)
#endif
int launch_sweep(gipuma_hip_session *s, int iteration, int colour, unsigned stages)
{
    const int gx = (s->cols + pm::kTileW - 1) / pm::kTileW;
    const int gy = (s->rows + pm::kSweepTileH - 1) / pm::kSweepTileH;
    const uint32_t phase = 1u + 2u * (uint32_t)iteration + (uint32_t)colour;
    sweep_fn k = pick_sweep(s);
    unsigned tune = s->tune | (s->costs_trusted ? 0u : Tune::kUntrustedCosts);
    // history rule (exact skipping (H) in pm_device.h): only inside a strictly alternating sequence of
    // full half-sweeps on trusted costs, as gipuma_hip_solve produces from its second iteration on
    const bool qualifies = stages == GIPUMA_STAGE_ALL && !s->unfused && s->costs_trusted;
    if (qualifies && s->prev1 == 1 - colour && s->prev2 == colour && !(tune & (Tune::kNoHistory | Tune::kNoSkip)))
        tune |= Tune::kHistorySkip;
    // will rule (H) hold for the next half-sweep if it is the other colour's full one?  (then prev1 = colour,
    // prev2 = today's prev1)
    const bool hist_next = qualifies && s->prev1 == 1 - colour && !(tune & (Tune::kNoHistory | Tune::kNoSkip));
    s->prev2 = s->prev1;
    s->prev1 = qualifies ? colour : -1;
    // push propagation: this half-sweep reads the costs of its propagation candidates from push_cost
    // (written by push_kernel after the previous half-sweep, or right now if nobody did), and offers
    // its own planes to the next one
    const int half_sweep = 2 * iteration + colour;
    const bool push_now = s->push_ok && qualifies && half_sweep < s->push_launches &&
                          !(tune & Tune::kNoSkip);
    if (push_now) {
        const bool hist = (tune & Tune::kHistorySkip) != 0;
        if (s->push_valid != colour || s->push_hist != hist) {
            const int rc = launch_push(s, 1 - colour, hist);
            if (rc) return rc;
        }
        tune |= Tune::kPushConsume;
        s->n_push_consumed++;
    }
    // plane-keyed propagation for the later half-sweeps (any skip rule the sweep would apply is applied there)
    bool fused_group = false;
    if (!push_now && s->group_ok && s->group_fused && qualifies && half_sweep >= s->group_from && !(tune & Tune::kNoSkip)) {
        fused_group = true;
    } else if (!push_now && s->group_ok && qualifies && half_sweep >= s->group_from && !(tune & Tune::kNoSkip)) {
        const int th = s->timed_half_sweep;
        const bool timed = th >= 0 && (size_t)(2 * th + 1) < s->gev.size();
        if (timed) HIP_OK(hipEventRecord(s->gev[2 * th], s->stream));
        const int rc = launch_group(s, colour, (tune & Tune::kHistorySkip) != 0, tune);
        if (rc) return rc;
        if (timed) {
            HIP_OK(hipEventRecord(s->gev[2 * th + 1], s->stream));
            s->gev_used[th] = 1;
        }
        tune |= Tune::kPushConsume;
    }
    s->push_valid = -1;  // the planes of `colour` are about to change
    const bool push_next = s->push_ok && qualifies && half_sweep + 1 < s->push_launches &&
                           !(tune & Tune::kNoSkip);
    // task order (performance only): planes are still incoherent in the first two iterations, where
    // grouping the evaluations of one plane saves cache-line fills; afterwards owner order is faster
    if (iteration >= 2 && !(tune & Tune::kSourceMajorTasks)) tune |= Tune::kOwnerMajorTasks;
    // ... and in those iterations the evaluations themselves are done column-per-lane (8 lanes per
    // (pixel, plane) pair, pm::sweep_cols_kernel) when the problem has that instantiation
    // (box 15 only: its 8 window columns fill the 8 lanes of a group; box 11, 6 of 8 lanes, measured
    // slower than one lane per pixel on config B: 18.0 vs 19.8 Mpix/s)
    const bool cols_ok = s->u8 && ((s->ch == 1 && s->hp.magic_addr && (s->box == 15 || s->box == 19 || s->box == 25)) ||
                                   (s->ch == 4 && s->box == 15)) &&
                         !(tune & (Tune::kNoColsKernel | Tune::kNoInterior));
    size_t lds = s->lds_sweep;
    // measured: box 15 (groups of 8 lanes) wins the first four half-sweeps of config C, box 25 (13 of 16
    // lanes) the first three of config D (128.7 / 90.9 / 73.7 -> 88.3 / 78.7 / 72.0 ms, the fourth loses)
    const int cols_launches = s->cols_launches >= 0 ? s->cols_launches : (s->box == 25 ? 3 : s->box == 19 ? 2 : 4);
    if (cols_ok && (2 * iteration + colour < cols_launches || (tune & Tune::kColsAlways))) {
        if (s->ch == 4)
            k = s->combine_reg ? pm::sweep_cols_kernel<15, true, 4> : pm::sweep_cols_kernel<15, false, 4>;
        else if (s->box == 15)
            k = s->combine_reg ? pm::sweep_cols_kernel<15, true> : pm::sweep_cols_kernel<15, false>;
        else if (s->box == 19)
            k = s->combine_reg ? pm::sweep_cols_kernel<19, true> : pm::sweep_cols_kernel<19, false>;
        else
            k = s->combine_reg ? pm::sweep_cols_kernel<25, true> : pm::sweep_cols_kernel<25, false>;
    }
    if (s->worder && !s->worder_valid) {
        typedef void (*order_fn)(const pm::Problem *, uint32_t *);
        const order_fn ok = s->ch == 4    ? (s->box == 15   ? pm::weight_order_kernel<15, 4>
                                             : s->box == 25 ? pm::weight_order_kernel<25, 4>
                                                            : pm::weight_order_kernel<11, 4>)
                            : s->box == 15 ? pm::weight_order_kernel<15>
                            : s->box == 25 ? pm::weight_order_kernel<25>
                            : s->box == 19 ? pm::weight_order_kernel<19>
                                           : pm::weight_order_kernel<11>;
        const int n = s->rows * s->cols;
        hipLaunchKernelGGL(ok, dim3((n + pm::kThreads - 1) / pm::kThreads), dim3(pm::kThreads), 0, s->stream, s->dp,
                           s->worder);
        HIP_OK(hipGetLastError());
        s->worder_valid = true;
    }
    if (fused_group) {
        // propagation costs per plane + accept replay + refinement in one launch (pm_group.h)
        typedef void (*fused_fn)(const pm::Problem *, float4 *, float *, int, uint32_t, unsigned);
#ifdef PM_FUSED_COLOUR_EXPERIMENT
        const fused_fn fk = s->ch == 4 ? pm::sweep_group_kernel<15, 4> : s->box == 15 ? pm::sweep_group_kernel<15> : s->box == 25 ? pm::sweep_group_kernel<25> : pm::sweep_group_kernel<11>;
        const size_t glds = sizeof(float) * (size_t)(s->ch == 4     ? pm::GroupLayout<15, 4>::total
                                                     : s->box == 15 ? pm::GroupLayout<15>::total
                                                     : s->box == 25 ? pm::GroupLayout<25>::total
                                                                    : pm::GroupLayout<11>::total);
#else
        const fused_fn fk = s->box == 15   ? pm::sweep_group_kernel<15>
                            : s->box == 19 ? pm::sweep_group_kernel<19>
                            : s->box == 25 ? pm::sweep_group_kernel<25>
                                           : pm::sweep_group_kernel<11>;
        const size_t glds = sizeof(float) * (size_t)(s->box == 15   ? pm::GroupLayout<15>::total
                                                     : s->box == 19 ? pm::GroupLayout<19>::total
                                                     : s->box == 25 ? pm::GroupLayout<25>::total
                                                                    : pm::GroupLayout<11>::total);
#endif
        const size_t flds = std::max(glds, s->lds_sweep);
        if (s->tile_order) {  // this launch's dispatch order from the colour's previous durations (identity without any)
            hipLaunchKernelGGL(pm::tile_order_kernel, dim3(8), dim3(pm::kThreads), 0, s->stream, s->dp, colour, tune, s->tile_order);
            HIP_OK(hipGetLastError());
        }
        if (!s->fused_attr_set) {
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
            s->fused_attr_set = true;
        }
#ifdef PM_WG_TICKS
        if (s->wg_ticks) {
            std::vector<unsigned long long> init((size_t)4 * gx * gy, 0ull);
            for (size_t i = 2; i < init.size(); i += 4) init[i] = ~0ull;
            HIP_OK(hipMemcpy(s->wg_ticks, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
#endif
        hipLaunchKernelGGL(fk, dim3(gx * gy), dim3(pm::kThreads), flds, s->stream, s->dp, s->norm4, s->cost, colour,
                           phase, tune);
        HIP_OK(hipGetLastError());
#ifdef PM_WG_TICKS
        if (s->wg_ticks) {
            std::vector<unsigned long long> h((size_t)4 * gx * gy);
            HIP_OK(hipStreamSynchronize(s->stream));
            HIP_OK(hipMemcpy(h.data(), s->wg_ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(getenv("GIPUMA_HIP_WG_TICKS"), "ab")) {
                const unsigned long long hdr[4] = {(unsigned long long)gx, (unsigned long long)gy, (unsigned long long)phase, (unsigned long long)tune};
                fwrite(hdr, sizeof hdr, 1, f);
                fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
                fclose(f);
            }
        }
#endif
        return 0;
    }
    hipLaunchKernelGGL(k, dim3(gx * gy), dim3(pm::kThreads), lds, s->stream, s->dp, s->norm4,
                       s->cost, colour, phase, stages, tune);
    HIP_OK(hipGetLastError());
    if (push_next) return launch_push(s, colour, hist_next);
    return 0;
}

int launch_dense(gipuma_hip_session *s, bool generate, float4 *planes, float *cost_out)
{
    const int gx = (s->cols + pm::kTileW - 1) / pm::kTileW;
    const int gy = (s->rows + pm::kDenseTileH - 1) / pm::kDenseTileH;
    init_fn k = generate ? pick_init<true>(s) : pick_init<false>(s);
    // random (or arbitrary caller-supplied) planes: column-per-lane evaluation where it exists
    if (s->u8 && s->ch == 4 && s->box == 15 && !(s->tune & (Tune::kNoColsKernel | Tune::kNoInterior))) {
        k = generate ? pm::init_cols_kernel<15, true, 4> : pm::init_cols_kernel<15, false, 4>;
    } else if (s->u8 && s->ch == 1 && s->hp.magic_addr && (s->box == 15 || s->box == 19 || s->box == 25) &&
               !(s->tune & (Tune::kNoColsKernel | Tune::kNoInterior))) {
        if (s->box == 15)
            k = generate ? pm::init_cols_kernel<15, true> : pm::init_cols_kernel<15, false>;
        else if (s->box == 19)
            k = generate ? pm::init_cols_kernel<19, true> : pm::init_cols_kernel<19, false>;
        else
            k = generate ? pm::init_cols_kernel<25, true> : pm::init_cols_kernel<25, false>;
    }
    hipLaunchKernelGGL(k, dim3(gx * gy), dim3(pm::kThreads), s->lds_dense, s->stream, s->dp, planes,
                       cost_out, s->tune);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int gipuma_hip_version(void) { return GIPUMA_HIP_ABI_VERSION; }

const char *gipuma_hip_last_error(void) { return g_err.c_str(); }

int gipuma_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gipuma_hip_cache_clear(void)
{
#ifndef GIPUMA_HIP_FLAVOUR_TU
    for (const FlavourApi *api : {&kFastApi, &kLiteralApi})  // (the other flavours keep their own packed planes)
        if (const int rc = api->cache_clear()) {
            g_err = api->last_error();
            return rc;
        }
#endif
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    for (auto &kv : g_cache)
        if (kv.second.users > 0)
            return fail(GIPUMA_HIP_ERR_ARG, "gipuma_hip_cache_clear: a live session still reads a cached packed image; "
                                            "destroy the sessions first");
    for (auto &kv : g_cache) {
        if (kv.second.packed) {
            (void)hipSetDevice(std::get<0>(kv.first));
            (void)hipFree(kv.second.packed);
        }
    }
    g_cache.clear();
    return 0;
}

int gipuma_hip_selftest_reciprocal(int device_id, unsigned long long *mismatches)
{
    if (!mismatches) return fail(GIPUMA_HIP_ERR_ARG, "null argument");
    if (device_id < 0 || device_id >= gipuma_hip_device_count())
        return fail(GIPUMA_HIP_ERR_NO_DEVICE, "no such HIP device");
    HIP_OK(hipSetDevice(device_id));
    unsigned long long *d = nullptr;
    HIP_OK(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemset(d, 0, sizeof *d);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pm::rcp_selftest_kernel, dim3(65536), dim3(pm::kThreads), 0, 0, d, 1u, 252u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, d, sizeof *d, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GIPUMA_HIP_ERR_DEVICE, "selftest: %s", hipGetErrorString(e));
    return 0;
}

int gipuma_hip_selftest_quotient(int device_id, unsigned z_first, unsigned z_count, unsigned long long *mismatches)
{
    if (!mismatches) return fail(GIPUMA_HIP_ERR_ARG, "null argument");
    if (device_id < 0 || device_id >= gipuma_hip_device_count())
        return fail(GIPUMA_HIP_ERR_NO_DEVICE, "no such HIP device");
    if (z_first >= (1u << 23) || z_count > (1u << 23) - z_first) return fail(GIPUMA_HIP_ERR_ARG, "significand range out of 0..2^23");
    HIP_OK(hipSetDevice(device_id));
    unsigned long long *d = nullptr;
    HIP_OK(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemset(d, 0, sizeof *d);
    // (launches of at most 2^14 denominators: ~0.1 s each, so that no single launch runs for minutes)
    for (unsigned done = 0; e == hipSuccess && done < z_count; done += 1u << 14) {
        const unsigned n = std::min(z_count - done, 1u << 14);
        hipLaunchKernelGGL(pm::quotient_selftest_kernel, dim3(n), dim3(pm::kThreads), 0, 0, d, z_first + done);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, d, sizeof *d, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GIPUMA_HIP_ERR_DEVICE, "selftest: %s", hipGetErrorString(e));
    return 0;
}

int gipuma_hip_create(const gipuma_hip_desc *d, gipuma_hip_session **out)
{
    if (!out) return fail(GIPUMA_HIP_ERR_ARG, "null out pointer");
    *out = nullptr;
    int rc = validate(d);
    if (rc) return rc;
    if (gipuma_hip_device_count() < 1)
        return fail(GIPUMA_HIP_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (d->device_id < 0 || d->device_id >= gipuma_hip_device_count())
        return fail(GIPUMA_HIP_ERR_ARG, "device_id out of range");
    gipuma_hip_session *s = new (std::nothrow) gipuma_hip_session;
    if (!s) return fail(GIPUMA_HIP_ERR_DEVICE, "out of host memory");
#ifndef GIPUMA_HIP_FLAVOUR_TU
    if (d->flags & (GIPUMA_HIP_FLAG_FAST | GIPUMA_HIP_FLAG_LITERAL)) {  // another flavour: this object is only its handle
        if ((d->flags & GIPUMA_HIP_FLAG_FAST) && (d->flags & GIPUMA_HIP_FLAG_LITERAL)) {
            delete s;
            return fail(GIPUMA_HIP_ERR_ARG, "GIPUMA_HIP_FLAG_FAST and GIPUMA_HIP_FLAG_LITERAL exclude each other");
        }
        s->api = (d->flags & GIPUMA_HIP_FLAG_LITERAL) ? &kLiteralApi : &kFastApi;
        rc = s->api->create(d, &s->impl);
        if (rc) {
            g_err = s->api->last_error();
            delete s;
            return rc;
        }
        *out = s;
        return 0;
    }
#endif

    // from here on, destroy() cleans up whatever was built.  The image cache is locked while this call looks at /
    // adds entries; a failure inside that region first takes back the packed planes this call put into the cache
    // (never verified), then UNLOCKS -- destroy() takes the same non-recursive mutex to give the use counts back.
    std::unique_lock<std::mutex> cache_lock(g_cache_mutex, std::defer_lock);
    std::vector<CachedImage *> fresh_cached;  // cache entries whose `packed` this call allocated
    auto abandon = [&]() {
        std::string keep = g_err;
        for (CachedImage *e : fresh_cached) {
            if (e->packed) (void)hipFree(e->packed);
            e->packed = nullptr;
        }
        fresh_cached.clear();
        if (cache_lock.owns_lock()) cache_lock.unlock();
        gipuma_hip_destroy(s);
        g_err = keep;
    };
#define CREATE_OK(expr)                        \
    do {                                       \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) {                \
            fail(GIPUMA_HIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
            abandon();                         \
            return GIPUMA_HIP_ERR_DEVICE;      \
        }                                      \
    } while (0)
    s->device = d->device_id;
    CREATE_OK(hipSetDevice(s->device));
    s->rows = d->rows;
    s->cols = d->cols;
    s->n_sel = d->n_selected;
    s->iterations = d->params.iterations;
    s->unfused = (d->flags & GIPUMA_HIP_FLAG_UNFUSED) != 0;
    if (const char *t = exp_env("TUNE")) {
        s->tune = (unsigned)strtoul(t, nullptr, 0);
        s->tune &= ~(Tune::kHistorySkip | Tune::kUntrustedCosts | Tune::kAccumChanged | Tune::kPushConsume);  // host-internal bits
    }
    if (const char *t = exp_env("COLS_LAUNCHES")) s->cols_launches = atoi(t);  // experiment
    if (const char *t = exp_env("LAUNCH_TIMES")) s->launch_times = atoi(t) != 0;
    if (d->stream) {
        s->stream = (hipStream_t)d->stream;
    } else {
        CREATE_OK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        s->own_stream = true;
    }
    for (auto &e : s->ev) CREATE_OK(hipEventCreate(&e));

    const bool on_device = (d->flags & GIPUMA_HIP_FLAG_IMAGES_ON_DEVICE) != 0;
    const size_t np = (size_t)d->rows * (size_t)d->cols;
    pm::Problem &hp = s->hp;
    hp.rows = d->rows;
    hp.cols = d->cols;
    s->ch = d->channels;
    hp.channels = d->channels;
    hp.pitch = on_device ? d->pitch : d->cols * d->channels;
    hp.n_sel = d->n_selected;
    hp.box_h = d->params.box_hsize;
    hp.box_v = d->params.box_vsize;
    hp.n_best = d->params.n_best;
    hp.cost_comb = d->params.cost_comb;
    hp.alpha = d->params.alpha;
    hp.tau_color = d->params.tau_color;
    hp.tau_gradient = d->params.tau_gradient;
    hp.gamma = d->params.gamma;
    hp.min_disp = d->params.min_disparity;
    hp.max_disp = d->params.max_disparity;
    hp.good_factor = d->params.good_factor;
    hp.seed = d->seed;

    // images: bind resident planes, or upload the reference + the selected views (compact pitch)
    auto resident = [&](int idx, const float **dst) -> hipError_t {
        if (on_device) {
            *dst = d->images[idx];
            return hipSuccess;
        }
        float *p = nullptr;
        const size_t row_bytes = (size_t)d->cols * d->channels * sizeof(float);
        hipError_t e = hipMalloc(&p, row_bytes * d->rows);
        if (e != hipSuccess) return e;
        s->owned.push_back(p);
        *dst = p;
        return hipMemcpy2DAsync(p, row_bytes, d->images[idx], (size_t)d->pitch * sizeof(float), row_bytes,
                                (size_t)d->rows, hipMemcpyHostToDevice, s->stream);
    };
    CREATE_OK(resident(0, &hp.ref.raw));
    for (int i = 0; i < d->n_selected; i++) CREATE_OK(resident(d->selected[i], &hp.view[i].img.raw));

    // U8 mode (weight table + window-packed source views) if every image handed to the path is
    // integer valued in [0,255] -- 8-bit input converted to float, main.cpp:941
    {
        const bool cached = on_device && (d->flags & GIPUMA_HIP_FLAG_CACHE_IMAGES) != 0;
        if (cached) cache_lock.lock();
        auto entry = [&](const float *img) -> CachedImage * {
            return cached ? &g_cache[CacheKey(s->device, img, d->rows, d->cols, hp.pitch, d->channels)] : nullptr;
        };
        // one flag per checked plane, planes whose verdict is cached are skipped
        const int n_planes = 1 + d->n_selected;
        CREATE_OK(hipMalloc(&s->flag, sizeof(int) * n_planes));
        CREATE_OK(hipMemsetAsync(s->flag, 0, sizeof(int) * n_planes, s->stream));
        const dim3 cg((d->cols + pm::kThreads - 1) / pm::kThreads, d->rows);
        auto check = s->ch == 4 ? pm::check_u8_kernel_c4 : pm::check_u8_kernel;
        std::vector<int> verdict(n_planes, -1);
        for (int i = 0; i < n_planes; i++) {
            const float *img = i == 0 ? hp.ref : hp.view[i - 1].img;
            CachedImage *e = entry(img);
            if (e && e->not_u8 >= 0)
                verdict[i] = e->not_u8;
            else
                hipLaunchKernelGGL(check, cg, dim3(pm::kThreads), 0, s->stream, img, hp.rows, hp.cols, hp.pitch,
                                   s->flag + i);
        }
        CREATE_OK(hipGetLastError());
        std::vector<int> flags(n_planes, 1);
        CREATE_OK(hipMemcpyAsync(flags.data(), s->flag, sizeof(int) * n_planes, hipMemcpyDeviceToHost, s->stream));
        CREATE_OK(hipStreamSynchronize(s->stream));
        int not_u8 = 0;
        for (int i = 0; i < n_planes; i++) {
            if (verdict[i] < 0) {
                verdict[i] = flags[i] != 0;
                if (CachedImage *e = entry(i == 0 ? hp.ref : hp.view[i - 1].img)) e->not_u8 = verdict[i];
            }
            not_u8 |= verdict[i];
        }
        s->u8 = !not_u8 && !(s->tune & Tune::kNoLut);
        hp.pw = d->cols + 8;
        // float-encoded window offsets need every entry index of a gray packed plane below 2^21
        hp.magic_addr = s->u8 && s->ch == 1 && !(s->tune & Tune::kNoMagicAddr) &&
                        (size_t)(d->rows + 3) * hp.pw <= (size_t)pm::kMagicMaxWords;
        if (s->u8) {
            const size_t words = (size_t)(d->rows + 3) * hp.pw * (s->ch == 4 ? 3 : 1);
            auto pack = s->ch == 4 ? pm::pack_kernel_c4 : pm::pack_kernel;
            const dim3 pgid((hp.pw + pm::kThreads - 1) / pm::kThreads, d->rows + 3);
            for (int i = 0; i < d->n_selected; i++) {
                CachedImage *e = entry(hp.view[i].img);
                if (e) {  // (counted once per use: destroy gives every one back)
                    e->users++;
                    s->cache_refs.push_back(CacheKey(s->device, hp.view[i].img, d->rows, d->cols, hp.pitch, d->channels));
                }
                if (e && e->packed) {  // packed for an earlier session: shared, owned by the cache
                    hp.view[i].packed = e->packed;
                    continue;
                }
                uint32_t *pk = nullptr;
                CREATE_OK(hipMalloc(&pk, words * sizeof(uint32_t)));
                if (e) {
                    e->packed = pk;
                    fresh_cached.push_back(e);
                } else {
                    s->packed.push_back(pk);
                }
                hp.view[i].packed = pk;
                hipLaunchKernelGGL(pack, pgid, dim3(pm::kThreads), 0, s->stream, hp.view[i].img,
                                   hp.rows, hp.cols, hp.pitch, hp.pw, pk);
            }
            CREATE_OK(hipGetLastError());
            if (cached) CREATE_OK(hipStreamSynchronize(s->stream));  // other sessions' streams may read them next
        }
        fresh_cached.clear();  // packed and synchronised: they belong to the cache now
        if (cache_lock.owns_lock()) cache_lock.unlock();
    }

    // cameras -> one POD block
    const gipuma_hip_camera &c0 = d->cameras[0];
    copy9(hp.rc.K_inv, c0.K_inv);
    copy9(hp.rc.M_inv, c0.M_inv);
    copy9(hp.rc.R_orig_inv, c0.R_orig_inv);
    copy3(hp.rc.P_col34, c0.P_col34);
    copy3(hp.rc.C, c0.C);
    hp.rc.fx = c0.fx;
    hp.rc.cx = c0.K[2];  // cam.K[2], cam.K[2+3] in getDepthFromPlane3_cu, gipuma.cu:699-701
    hp.rc.cy = c0.K[5];
    hp.rc.alpha = c0.alpha;
    hp.rc.f = c0.f;
    hp.rc.baseline = c0.baseline;
    hp.rc.depth_min = c0.depth_min;
    hp.rc.depth_max = c0.depth_max;
    for (int i = 0; i < d->n_selected; i++) {
        const gipuma_hip_camera &c = d->cameras[d->selected[i]];
        copy9(hp.view[i].K, c.K);
        copy9(hp.view[i].R, c.R);
        copy3(hp.view[i].t, c.t);
#if PM_APPROX && defined(PM_APPROX_HFOLD)
        // A = K R K_ref^-1, u = K t (homography(), approx flavour), formed in double
        double KR[9];
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++)
                KR[3 * r + q] = (double)c.K[3 * r] * c.R[q] + (double)c.K[3 * r + 1] * c.R[3 + q] + (double)c.K[3 * r + 2] * c.R[6 + q];
        for (int r = 0; r < 3; r++) {
            for (int q = 0; q < 3; q++)
                hp.view[i].A[3 * r + q] = (float)(KR[3 * r] * c0.K_inv[q] + KR[3 * r + 1] * c0.K_inv[3 + q] + KR[3 * r + 2] * c0.K_inv[6 + q]);
            hp.view[i].u[r] = (float)((double)c.K[3 * r] * c.t[0] + (double)c.K[3 * r + 1] * c.t[1] + (double)c.K[3 * r + 2] * c.t[2]);
        }
#endif
    }
    {
        const size_t tiles = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) *
                             (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        s->et_hint_bytes = tiles * 12;
        CREATE_OK(hipMalloc(&s->et_hint, s->et_hint_bytes));
        CREATE_OK(hipMemsetAsync(s->et_hint, 0, s->et_hint_bytes, s->stream));
        hp.et_hint = s->et_hint;
        CREATE_OK(hipMalloc(&s->et_stat, 3 * pm::kEtSlot * sizeof(unsigned)));
        CREATE_OK(hipMemsetAsync(s->et_stat, 0, 3 * pm::kEtSlot * sizeof(unsigned), s->stream));
        hp.et_stat = s->et_stat;
    }
    // Skip rule (S) -- a ring of the last 8 planes a pixel's propagation evaluated, 129 B per pixel -- only where a
    // propagation candidate is expensive and nothing else shares its evaluation: colour sessions (their images are four
    // times the gray ones; measured with the round-3 library: late half-sweeps 6-9 % fewer tasks).  Gray sessions run
    // the plane-keyed propagation kernel instead and keep their footprint (config C: 0.3 % for 248 MB).
    // Performance only: without the memory the rule is off.
    if (s->ch == 4 && !(s->tune & (Tune::kNoSeen | Tune::kNoSkip))) {
        if (hipMalloc(&s->seen_ring, (size_t)pm::kSeenRing * np * sizeof(float4)) == hipSuccess &&
            hipMalloc(&s->seen_pos, np) == hipSuccess) {
            CREATE_OK(hipMemsetAsync(s->seen_pos, 0, np, s->stream));
            hp.seen_ring = s->seen_ring;
            hp.seen_pos = s->seen_pos;
        } else {
            (void)hipGetLastError();
            if (s->seen_ring) (void)hipFree(s->seen_ring);
            s->seen_ring = nullptr;
            s->seen_pos = nullptr;
        }
    }
#ifdef PM_CHECKED  // (the bounds-checked TEST build, pm_core.h)
    CREATE_OK(hipMalloc(&s->viol, pm::kDbgSlots * sizeof(unsigned long long)));
    CREATE_OK(hipMemsetAsync(s->viol, 0, pm::kDbgSlots * sizeof(unsigned long long), s->stream));
    hp.viol = s->viol;
#endif
    if (exp_env("COUNTS") && atoi(exp_env("COUNTS"))) {  // experiment aid
        CREATE_OK(hipMalloc(&s->dbg, 64 * pm::kDbgSlots * sizeof(unsigned long long)));
        CREATE_OK(hipMemsetAsync(s->dbg, 0, 64 * pm::kDbgSlots * sizeof(unsigned long long), s->stream));
        hp.dbg = s->dbg;
    }
#ifdef PM_WG_TICKS
    if (getenv("GIPUMA_HIP_WG_TICKS")) {
        const size_t sweep_tiles_early = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) * (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        CREATE_OK(hipMalloc(&s->wg_ticks, 4 * sweep_tiles_early * sizeof(unsigned long long)));
        hp.wg_ticks = s->wg_ticks;
    }
#endif
    CREATE_OK(hipMalloc(&s->changed, np));
    CREATE_OK(hipMemsetAsync(s->changed, 1, np, s->stream));
    hp.changed = s->changed;

    // state planes, zero-filled like LineState::resize (linestate.h:16-24)
    CREATE_OK(hipMalloc(&s->norm4, np * sizeof(float4)));
    CREATE_OK(hipMalloc(&s->cost, np * sizeof(float)));
    CREATE_OK(hipMemsetAsync(s->norm4, 0, np * sizeof(float4), s->stream));
    CREATE_OK(hipMemsetAsync(s->cost, 0, np * sizeof(float), s->stream));

    // kernel variant
    s->box = 0;
    if (hp.box_h == hp.box_v && !(s->tune & Tune::kGenericBox) &&
        (hp.box_h == 11 || hp.box_h == 15 || hp.box_h == 25 || (hp.box_h == 19 && s->ch == 1)))
        s->box = hp.box_h;
    {
        // the specialised loops fold the gradient term's 1/16 into alpha and tau_gradient (dis_fold, pm_cost.h): exact
        // unless alpha / 16 is subnormal or 16 tau_gradient overflows -- such parameters take the literal generic loop
        const float a16 = hp.alpha * 0.0625f, tg16 = hp.tau_gradient * 16.0f;
        const bool fold_exact = a16 * 16.0f == hp.alpha && (std::isfinite(tg16) || !std::isfinite(hp.tau_gradient));
        if (!fold_exact) s->box = 0;
    }
    s->combine_reg = hp.cost_comb == GIPUMA_COMB_BEST_N && hp.n_best >= 1 && hp.n_best <= 4 &&
                     !(s->tune & Tune::kGenericCombine);
    if ((s->tune & Tune::kNoInterior) && !(s->box == 15 && s->u8 && s->combine_reg)) {
        s->box = 0;  // the no-interior A/B arm only exists for these two variants
        s->combine_reg = false;
    }
    // early termination of refinement evaluations (pm::multiview_cost): only where every view cost is
    // provably finite and below MAXCOST for every plane, so that numValid == n_sel always
    // (gipuma.cu:771-775): weights exp(-k/gamma) <= 1 from the table, dis <= (1-alpha)*tau_c + alpha*tau_g
    {
        const gipuma_hip_params &p = d->params;
        const double samples = (double)((hp.box_h + 1) / 2) * (double)((hp.box_v + 1) / 2);
        const bool sane = p.gamma > 0.0f && p.alpha >= 0.0f && p.alpha <= 1.0f && p.tau_color >= 0.0f &&
                          p.tau_gradient >= 0.0f && std::isfinite(p.tau_color) && std::isfinite(p.tau_gradient) &&
                          samples * ((1.0 - p.alpha) * p.tau_color + (double)p.alpha * p.tau_gradient) * 1.01 <
                              (double)GIPUMA_HIP_MAXCOST;
        // ... and only where a half-sweep is many waves of workgroups: on a frame whose tiles all fit the
        // GPU at once (< 1024 = 256 CUs x 4) the launch lasts as long as its slowest workgroup, and
        // the occasional redo pass of a bounded evaluation lengthens exactly that (configs A, B: -5..-13 %)
        const size_t tiles = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) *
                             (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        const bool big = tiles >= 1024 || exp_env("ET_FORCE") != nullptr;  // (env: tests on small frames)
        // (gray: the pipelined loop on float-encoded offsets; colour: its integer-addressed loop)
        hp.et_enable = sane && big && s->u8 && s->combine_reg && (s->ch == 4 || (hp.magic_addr && s->box > 0));
        // GIPUMA_HIP_ET_FORCE=2 (tests): every workgroup bounds every step, whatever the probes measured
        if (hp.et_enable && exp_env("ET_FORCE") && atoi(exp_env("ET_FORCE")) >= 2) hp.et_enable = 2;
        hp.et_theta[0] = 1.0f;
        hp.et_theta[1] = 1.0f;
        // the two-phase refinement (compile-time box) redoes open candidates item by item, which
        // is cheap; the per-wavefront bound repeats the whole wavefront and wants a looser third bound
        const bool two_phase = s->box > 0 && !(s->tune & Tune::kNoTwoPhase);
        hp.et_theta[2] = two_phase ? 1.0f : 1.5f;
        if (const char *g = exp_env("TP_G0")) hp.tp_g0 = atoi(g);  // experiment: phase-1 columns
        if (const char *t = exp_env("ET_THETA")) {  // experiment: "t0,t1,t2" (any value is exact)
            float a, b, c;
            if (sscanf(t, "%f,%f,%f", &a, &b, &c) == 3) {
                hp.et_theta[0] = a;
                hp.et_theta[1] = b;
                hp.et_theta[2] = c;
            }
        }
    }
    // lower-bound prefilter of refinement candidates: where the two-phase refinement runs on gray planes
    hp.lb_k = 0;  // chosen by the probe workgroups
    if (const char *t = exp_env("LB_K")) hp.lb_k = atoi(t);  // experiment: fixed length, < 0 = off
    if (hp.et_enable && s->box > 0 && hp.lb_k >= 0 && !(s->tune & (Tune::kNoTwoPhase | Tune::kNoEarlyExit))) {
        // (one plane of rows*cols words per two listed samples: 8 planes for box 15, 16 for box 25, 4 for box 11)
        const int lb_planes = (s->box == 15 ? pm::lb_max<15>() : s->box == 25 ? pm::lb_max<25>() : s->box == 19 ? pm::lb_max<19>() : pm::lb_max<11>()) / 2;
        // performance-only state: without the memory for it the solve runs without the prefilter, same results
        if (hipMalloc(&s->worder, (size_t)lb_planes * np * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            s->worder = nullptr;
        }
        hp.worder = s->worder;
    }
    if (!s->worder) hp.lb_k = -1;
    // push propagation (pm_push.h): box 11 / 15 / 25, register combiner, packed gray planes with float-encoded offsets
    // ... or colour (three words per texel, integer addressing), box 15
    s->push_ok = s->u8 && s->combine_reg && s->n_sel > 0 && !(s->tune & (Tune::kNoInterior | Tune::kNoSkip)) &&
                 ((s->ch == 1 && hp.magic_addr && (s->box == 11 || s->box == 15 || s->box == 19 || s->box == 25)) ||
                  (s->ch == 4 && s->box == 15));
    // measured: config C 4 (5 and 6 level), config D 3 (4 level, 6 loses), config B 2 (+1 %)
    // colour (config C geometry): 3 where the plane-keyed kernel takes over afterwards (frames of >= 1024 tiles:
    // 2 / 3 / 4 / 6 pushed half-sweeps 195.5 / 195.7 / 197.7 / 205.9 ms per view), else 6 (4: -1.3 %, 8: -0.7 %, 16: -7 %)
    const size_t sweep_tiles = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) *
                               (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
    s->push_launches = s->ch == 4 ? (sweep_tiles >= 1024 ? 3 : 6) : s->box == 15 ? 4 : s->box == 25 ? 3 : 2;  // (box 19: 2 / 3 / 4 -> 131.7 / 134.4 / 139.0 ms)
    if (const char *t = exp_env("PUSH_LAUNCHES")) s->push_launches = atoi(t);  // A/B runs: 0 = never
    // plane-keyed propagation (pm_group.h) after the pushed half-sweeps.  The kernels exist for boxes 11 / 15 / 25 in gray and
    // box 15 in colour; the DEFAULT schedule uses them for boxes 15 and 25 (gray) and box 15 (colour) on frames of >= 1024
    // tiles.  Box 11 and every frame under 1024 tiles (configs A and B) keep group_from = -1: their instantiation is
    // reached only through GIPUMA_HIP_GROUP_FROM under GIPUMA_HIP_EXPERIMENTS (and is parity-tested there).
    s->group_ok = s->push_ok && ((s->ch == 1 && (s->box == 11 || s->box == 15 || s->box == 19 || s->box == 25)) || (s->ch == 4 && s->box == 15));
    if (s->push_launches <= 0) s->push_ok = false;
    // Default: right after the pushed half-sweeps -- from the fifth half-sweep on for box 15 (config C 90.6 -> 80.8 ms per
    // view in round 4; any start between the third and the fifth within 0.5 %), from the fourth for box 25 and colour; on
    // config B's 300 tiles (one wave of workgroups) it loses 1.5 % (scripts/history/gpu_r04_sched.sh).
    // GIPUMA_HIP_GROUP_FROM=<first half-sweep> (experiments): < 0 = never.
    {
        const size_t tiles = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) *
                             (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        s->group_from = tiles < 1024 ? -1 : s->ch == 4 ? 3 : s->box == 15 ? 4 : s->box == 25 ? 3 : s->box == 19 ? 2 : -1;
    }
    if (const char *t = exp_env("GROUP_FROM")) s->group_from = atoi(t);
    // gray: ONE launch per half-sweep (pm::sweep_group_kernel).  Colour: pm::group_kernel<15, 4> in front of the sweep
    // kernel, two launches -- a fused colour instantiation is not built (it held two workgroups per CU at
    // 256 registers with 121 spilled, was slower, and could not be trusted).
    s->group_fused = s->ch == 1;
#ifdef PM_FUSED_COLOUR_EXPERIMENT
    if (const char *t = exp_env("GROUP_FUSED")) s->group_fused = atoi(t) != 0;  // (hunt builds: colour too)
#else
    if (const char *t = exp_env("GROUP_FUSED")) s->group_fused = s->ch == 1 && atoi(t) != 0;  // 0: group_kernel + sweep_kernel, two launches
#endif
    if (s->group_from < 0) s->group_ok = false;
    if (s->push_ok || s->group_ok) {
        // performance-only state too: without it every half-sweep evaluates its own propagation candidates
        if (hipMalloc(&s->push_cost, 8 * np * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            s->push_cost = nullptr;
            s->push_ok = s->group_ok = false;
        }
        hp.push_cost = s->push_cost;
    }
    s->lds_sweep = lds_bytes(s, pm::kSweepTileH, !s->combine_reg, true);
    s->lds_dense = lds_bytes(s, pm::kDenseTileH, true, false);
    if (s->lds_sweep > 160u * 1024u || s->lds_dense > 160u * 1024u) {  // 160 KiB of LDS per CU on gfx950
        fail(GIPUMA_HIP_ERR_UNSUPPORTED, "window x views needs more than 160 KiB of LDS per workgroup");
        abandon();
        return GIPUMA_HIP_ERR_UNSUPPORTED;
    }
    // dispatch order of the fused launches (pm::tile_order_kernel; performance-only state: without it the plain order).
    // GIPUMA_HIP_TILE_ORDER=0/1 under GIPUMA_HIP_EXPERIMENTS: A/B runs
    {
        bool want = PM_TILE_ORDER_DEFAULT != 0;
        if (const char *t = exp_env("TILE_ORDER")) want = atoi(t) != 0;
        const size_t tiles = (size_t)((d->cols + pm::kTileW - 1) / pm::kTileW) * (size_t)((d->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        if (want && s->group_ok && s->group_fused && tiles >= 8 && !(s->tune & Tune::kNoXcdRemap)) {
            if (hipMalloc(&s->tile_clock, 4 * tiles * sizeof(unsigned long long)) == hipSuccess &&
                hipMalloc(&s->tile_order, tiles * sizeof(int)) == hipSuccess) {
                CREATE_OK(hipMemsetAsync(s->tile_clock, 0, 4 * tiles * sizeof(unsigned long long), s->stream));
                hp.tile_clock = s->tile_clock;
                hp.tile_order = s->tile_order;
            } else {
                (void)hipGetLastError();
                if (s->tile_clock) (void)hipFree(s->tile_clock);
                s->tile_clock = nullptr;
                s->tile_order = nullptr;
            }
        }
    }
    CREATE_OK(hipMalloc(&s->dp, sizeof(pm::Problem)));
    CREATE_OK(hipMemcpyAsync(s->dp, &hp, sizeof(pm::Problem), hipMemcpyHostToDevice, s->stream));
    CREATE_OK(hipStreamSynchronize(s->stream));  // host image buffers may be released by the caller
#undef CREATE_OK
    *out = s;
    return 0;
}

#ifdef PM_CHECKED
// The bounds-checked TEST build (pm_core.h, -DPM_CHECKED): what the kernels of this session counted -- one line per session,
// appended to the file GIPUMA_CHECKED_LOG names (stderr without it): the accesses of each class that fell outside their
// buffer (window loads gray / integer-addressed / colour, norm4, cost, pushed costs, flags and rings).
static void checked_collect(gipuma_hip_session *s)
{
    unsigned long long h[pm::kDbgSlots] = {};
    if (hipMemcpy(h, s->viol, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long total = 0;
    for (unsigned long long v : h) total += v;
    const char *path = getenv("GIPUMA_CHECKED_LOG");
    FILE *f = path ? fopen(path, "a") : stderr;
    if (!f) f = stderr;
    fprintf(f, "gipuma_hip CHECKED session %dx%d ch %d box %d views %d: violations %llu (window %llu, window-int %llu, window-c4 %llu, "
               "norm4 %llu, cost %llu, push_cost %llu, flags %llu)\n", s->cols, s->rows, s->ch, s->box, s->n_sel, total, h[0], h[1], h[2],
            h[3], h[4], h[5], h[6]);
    if (f != stderr) fclose(f);
}
#endif

int gipuma_hip_destroy(gipuma_hip_session *s)
{
    if (!s) return 0;
#ifndef GIPUMA_HIP_FLAVOUR_TU
    if (s->api) {
        const int rc = s->api->destroy(s->impl);
        delete s;
        return rc;
    }
#endif
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (!s->cache_refs.empty()) {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        for (const CacheKey &k : s->cache_refs) {
            auto it = g_cache.find(k);
            if (it != g_cache.end() && it->second.users > 0) it->second.users--;
        }
        s->cache_refs.clear();
    }
    for (float *p : s->owned) (void)hipFree(p);
    for (uint32_t *p : s->packed) (void)hipFree(p);
    if (s->flag) (void)hipFree(s->flag);
    if (s->dp) (void)hipFree(s->dp);
    if (s->changed) (void)hipFree(s->changed);
    if (s->push_cost) (void)hipFree(s->push_cost);
    if (s->et_hint) (void)hipFree(s->et_hint);
    if (s->worder) (void)hipFree(s->worder);
#ifdef PM_CHECKED
    if (s->viol) {
        checked_collect(s);
        (void)hipFree(s->viol);
    }
#endif
    if (s->dbg) (void)hipFree(s->dbg);
    if (s->tile_clock) (void)hipFree(s->tile_clock);
    if (s->tile_order) (void)hipFree(s->tile_order);
    if (s->seen_ring) (void)hipFree(s->seen_ring);
    if (s->seen_pos) (void)hipFree(s->seen_pos);
    if (s->et_stat) (void)hipFree(s->et_stat);
    if (s->norm4) (void)hipFree(s->norm4);
    if (s->cost) (void)hipFree(s->cost);
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : s->lev) (void)hipEventDestroy(e);
    for (auto &e : s->gev) (void)hipEventDestroy(e);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return 0;
}

int gipuma_hip_init_planes(gipuma_hip_session *s)
{
    FORWARD(s, init_planes);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    HIP_OK(hipSetDevice(s->device));
    // (a fresh solve starts with fresh hints, so that repeated solves of a session do the same work)
    HIP_OK(hipMemsetAsync(s->et_hint, 0, s->et_hint_bytes, s->stream));
    HIP_OK(hipMemsetAsync(s->et_stat, 0, 3 * pm::kEtSlot * sizeof(unsigned), s->stream));
    s->worder_valid = false;  // (listed again by the first sweep: part of every solve)
    if (s->tile_clock) {  // (no durations yet: the first fused launch of either colour runs in the plain order)
        const size_t tiles = (size_t)((s->cols + pm::kTileW - 1) / pm::kTileW) * (size_t)((s->rows + pm::kSweepTileH - 1) / pm::kSweepTileH);
        HIP_OK(hipMemsetAsync(s->tile_clock, 0, 4 * tiles * sizeof(unsigned long long), s->stream));
    }
    if (s->seen_pos) HIP_OK(hipMemsetAsync(s->seen_pos, 0, (size_t)s->rows * s->cols, s->stream));  // rule (S): new planes
    const int rc = launch_dense(s, true, s->norm4, s->cost);
    if (!rc) s->costs_trusted = true;
    s->finalized = false;
    s->prev1 = s->prev2 = -1;
    s->push_valid = -1;
    return rc;
}

int gipuma_hip_sweep(gipuma_hip_session *s, int iteration, int colour, unsigned stages)
{
    FORWARD(s, sweep, iteration, colour, stages);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    if (iteration < 0 || (colour != GIPUMA_BLACK && colour != GIPUMA_RED) || (stages & ~7u))
        return fail(GIPUMA_HIP_ERR_ARG, "bad iteration/colour/stages");
    if (s->finalized)
        return fail(GIPUMA_HIP_ERR_ARG, "the session holds finalized maps (world normal, depth), not planes: call "
                                        "gipuma_hip_init_planes or gipuma_hip_set_state before sweeping again");
    HIP_OK(hipSetDevice(s->device));
    if (s->unfused) {  // the reference's three launches per colour, gipuma.cu:1915-1923
        for (unsigned st = 1; st <= 4; st <<= 1)
            if (stages & st) {
                int rc = launch_sweep(s, iteration, colour, st);
                if (rc) return rc;
            }
        return 0;
    }
    return stages ? launch_sweep(s, iteration, colour, stages) : 0;
}

int gipuma_hip_finalize(gipuma_hip_session *s)
{
    FORWARD(s, finalize);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    HIP_OK(hipSetDevice(s->device));
    const int n = s->rows * s->cols;
    if (s->finalized) return fail(GIPUMA_HIP_ERR_ARG, "already finalized");
    hipLaunchKernelGGL(pm::finalize_kernel, dim3((n + pm::kThreads - 1) / pm::kThreads), dim3(pm::kThreads), 0,
                       s->stream, s->dp, s->norm4, s->cost);
    HIP_OK(hipGetLastError());
    s->finalized = true;
    s->costs_trusted = false;
    s->prev1 = s->prev2 = -1;
    s->push_valid = -1;
    return 0;
}

int gipuma_hip_eval_cost(gipuma_hip_session *s, const float *planes_host, float *cost_out_host)
{
    FORWARD(s, eval_cost, planes_host, cost_out_host);
    if (!s || !planes_host || !cost_out_host) return fail(GIPUMA_HIP_ERR_ARG, "null argument");
    HIP_OK(hipSetDevice(s->device));
    const size_t np = (size_t)s->rows * (size_t)s->cols;
    float4 *pl = nullptr;
    float *c = nullptr;
    HIP_OK(hipMalloc(&pl, np * sizeof(float4)));
    hipError_t e = hipMalloc(&c, np * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(pl);
        return fail(GIPUMA_HIP_ERR_DEVICE, "hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = 0;
    e = hipMemcpyAsync(pl, planes_host, np * sizeof(float4), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) rc = launch_dense(s, false, pl, c);
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(cost_out_host, c, np * sizeof(float), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(s->stream);
    (void)hipFree(pl);
    (void)hipFree(c);
    if (e != hipSuccess) return fail(GIPUMA_HIP_ERR_DEVICE, "eval_cost: %s", hipGetErrorString(e));
    return rc;
}

int gipuma_hip_get_state(gipuma_hip_session *s, float *norm4_host, float *cost_host)
{
    FORWARD(s, get_state, norm4_host, cost_host);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    HIP_OK(hipSetDevice(s->device));
    const size_t np = (size_t)s->rows * (size_t)s->cols;
    if (norm4_host)
        HIP_OK(hipMemcpyAsync(norm4_host, s->norm4, np * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    if (cost_host)
        HIP_OK(hipMemcpyAsync(cost_host, s->cost, np * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    return 0;
}

int gipuma_hip_set_state(gipuma_hip_session *s, const float *norm4_host, const float *cost_host)
{
    FORWARD(s, set_state, norm4_host, cost_host);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    HIP_OK(hipSetDevice(s->device));
    const size_t np = (size_t)s->rows * (size_t)s->cols;
    if (norm4_host)
        HIP_OK(hipMemcpyAsync(s->norm4, norm4_host, np * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    if (cost_host)
        HIP_OK(hipMemcpyAsync(s->cost, cost_host, np * sizeof(float), hipMemcpyHostToDevice, s->stream));
    if (s->seen_pos) HIP_OK(hipMemsetAsync(s->seen_pos, 0, np, s->stream));  // rule (S): a cost may have gone up
    HIP_OK(hipStreamSynchronize(s->stream));
    s->costs_trusted = false;
    s->prev1 = s->prev2 = -1;
    s->push_valid = -1;
    if (norm4_host) s->finalized = false;
    return 0;
}

int gipuma_hip_state_device_ptrs(gipuma_hip_session *s, float **norm4_dev, float **cost_dev)
{
    FORWARD(s, state_device_ptrs, norm4_dev, cost_dev);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    if (norm4_dev) *norm4_dev = (float *)s->norm4;
    if (cost_dev) *cost_dev = s->cost;
    return 0;
}

int gipuma_hip_solve(gipuma_hip_session *s, gipuma_hip_timing *timing)
{
    FORWARD(s, solve, timing);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    HIP_OK(hipSetDevice(s->device));
    int rc;
    int launches = 0;
    HIP_OK(hipEventRecord(s->ev[0], s->stream));
    if ((rc = gipuma_hip_init_planes(s))) return rc;
    HIP_OK(hipEventRecord(s->ev[1], s->stream));
    const size_t n_lev = (timing || s->launch_times) ? (size_t)(2 * s->iterations + 1) : 0;
    while (s->lev.size() < n_lev) {
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        s->lev.push_back(e);
    }
    s->n_push_consumed = 0;
    if (n_lev) {
        while (s->gev.size() < 2 * (n_lev - 1)) {
            hipEvent_t e;
            HIP_OK(hipEventCreate(&e));
            s->gev.push_back(e);
        }
        s->gev_used.assign(n_lev - 1, 0);
        HIP_OK(hipEventRecord(s->lev[0], s->stream));
    }
    for (int it = 0; it < s->iterations; it++) {  // gipuma.cu:1911-1941
        s->timed_half_sweep = n_lev ? 2 * it : -1;
        rc = gipuma_hip_sweep(s, it, GIPUMA_BLACK, GIPUMA_STAGE_ALL);
        if (!rc && n_lev) HIP_OK(hipEventRecord(s->lev[2 * it + 1], s->stream));
        s->timed_half_sweep = n_lev ? 2 * it + 1 : -1;
        if (!rc) rc = gipuma_hip_sweep(s, it, GIPUMA_RED, GIPUMA_STAGE_ALL);
        s->timed_half_sweep = -1;
        if (rc) return rc;
        if (n_lev) HIP_OK(hipEventRecord(s->lev[2 * it + 2], s->stream));
        launches += s->unfused ? 6 : 2;
    }
    HIP_OK(hipEventRecord(s->ev[2], s->stream));
    if ((rc = gipuma_hip_finalize(s))) return rc;
    HIP_OK(hipEventRecord(s->ev[3], s->stream));
    if (timing) {
        HIP_OK(hipEventSynchronize(s->ev[3]));
        HIP_OK(hipEventElapsedTime(&timing->ms_init, s->ev[0], s->ev[1]));
        HIP_OK(hipEventElapsedTime(&timing->ms_sweeps, s->ev[1], s->ev[2]));
        HIP_OK(hipEventElapsedTime(&timing->ms_finalize, s->ev[2], s->ev[3]));
        HIP_OK(hipEventElapsedTime(&timing->ms_total, s->ev[0], s->ev[3]));
        timing->n_sweep_launches = launches;
        timing->ms_sweep_avg = launches ? timing->ms_sweeps / (float)launches : 0.0f;
    }
    if (n_lev) {
        HIP_OK(hipEventSynchronize(s->ev[3]));
        s->half_sweep_ms.assign(n_lev - 1, 0.0f);
        for (size_t i = 1; i < n_lev; i++) HIP_OK(hipEventElapsedTime(&s->half_sweep_ms[i - 1], s->lev[i - 1], s->lev[i]));
        s->group_ms.assign(n_lev - 1, 0.0f);
        for (size_t i = 0; i + 1 < n_lev; i++)
            if (s->gev_used[i]) HIP_OK(hipEventElapsedTime(&s->group_ms[i], s->gev[2 * i], s->gev[2 * i + 1]));
        s->n_pushed = s->n_push_consumed;
        if (s->launch_times) {
            fprintf(stderr, "gipuma_hip launch_ms:");
            for (float ms : s->half_sweep_ms) fprintf(stderr, " %.3f", ms);
            fprintf(stderr, "\n");
        }
        if (s->dbg) {  // per half-sweep: events per pixel of the colour
            std::vector<unsigned long long> h(64 * pm::kDbgSlots);
            HIP_OK(hipMemcpy(h.data(), s->dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            HIP_OK(hipMemset(s->dbg, 0, h.size() * sizeof(unsigned long long)));
            const double px = 0.5 * (double)s->rows * (double)s->cols;
            static const char *names[] = {"tasks/px", "-", "items/px", "open items/px", "redone cands/px", "cands/px"};
            if (h[62 * pm::kDbgSlots + 1]) {  // the plane-keyed kernels' phase clocks: 100 MHz ticks of each workgroup's first wavefront, summed
                fprintf(stderr, "gipuma_hip plane-keyed phase ticks (state, task list, grouping, batches, wait for the last batch, tile, "
                                "replay + refinement):");
                for (int k = 0; k < 7; k++) fprintf(stderr, " %llu", h[62 * pm::kDbgSlots + k]);
                fprintf(stderr, "\n");
            }
            if (h[61 * pm::kDbgSlots + 0]) {  // pm::group_kernel's batches, summed over the solve's launches
                const double nb = (double)h[61 * pm::kDbgSlots + 0], nt = (double)h[61 * pm::kDbgSlots + 7];
                fprintf(stderr, "gipuma_hip group_kernel batches: %.1f per tile; per batch %.1f strips, %.1f tasks, %.2f groups, "
                                "%.2f rows; per tile %.1f groups, %.1f tasks\n", nb / nt, h[61 * pm::kDbgSlots + 1] / nb,
                        h[61 * pm::kDbgSlots + 2] / nb, h[61 * pm::kDbgSlots + 3] / nb, h[61 * pm::kDbgSlots + 4] / nb,
                        h[61 * pm::kDbgSlots + 5] / nt, h[61 * pm::kDbgSlots + 6] / nt);
            }
            for (int k = 0; k < 6; k++) {
                fprintf(stderr, "gipuma_hip counts %s:", names[k]);
                for (int ph = 1; ph <= 2 * s->iterations && ph < 64; ph++)
                    fprintf(stderr, " %.3f", (double)h[(size_t)ph * pm::kDbgSlots + k] / px);
                fprintf(stderr, "\n");
            }
        }
    }
    return 0;
}

int gipuma_hip_launch_times(gipuma_hip_session *s, float *ms_half_sweep, int capacity, int *n_half_sweeps, int *n_pushed)
{
    FORWARD(s, launch_times, ms_half_sweep, capacity, n_half_sweeps, n_pushed);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    const int n = (int)s->half_sweep_ms.size();
    if (ms_half_sweep)
        for (int i = 0; i < n && i < capacity; i++) ms_half_sweep[i] = s->half_sweep_ms[i];
    if (n_half_sweeps) *n_half_sweeps = n;
    if (n_pushed) *n_pushed = s->n_pushed;
    return 0;
}

int gipuma_hip_schedule(gipuma_hip_session *s, int info[4])
{
    FORWARD(s, schedule, info);
    if (!s || !info) return fail(GIPUMA_HIP_ERR_ARG, "null argument");
    info[0] = s->push_ok ? s->push_launches : 0;
    info[1] = s->group_ok ? s->group_from : -1;
    info[2] = s->group_ok && s->group_fused ? 1 : 0;
    const bool cols_ok = s->u8 && ((s->ch == 1 && s->hp.magic_addr && (s->box == 15 || s->box == 19 || s->box == 25)) || (s->ch == 4 && s->box == 15)) &&
                         !(s->tune & (Tune::kNoColsKernel | Tune::kNoInterior));
    info[3] = cols_ok ? (s->cols_launches >= 0 ? s->cols_launches : (s->box == 25 ? 3 : s->box == 19 ? 2 : 4)) : 0;
    return 0;
}

int gipuma_hip_group_times(gipuma_hip_session *s, float *ms_group, int capacity, int *n_half_sweeps)
{
    FORWARD(s, group_times, ms_group, capacity, n_half_sweeps);
    if (!s) return fail(GIPUMA_HIP_ERR_ARG, "null session");
    const int n = (int)s->group_ms.size();
    if (ms_group)
        for (int i = 0; i < n && i < capacity; i++) ms_group[i] = s->group_ms[i];
    if (n_half_sweeps) *n_half_sweeps = n;
    return 0;
}

int gipuma_hip_run(const gipuma_hip_desc *desc, float *norm4_out, float *cost_out, gipuma_hip_timing *timing)
{
    gipuma_hip_session *s = nullptr;
    int rc = gipuma_hip_create(desc, &s);
    if (rc) return rc;
    gipuma_hip_timing t{};
    rc = gipuma_hip_solve(s, &t);
    if (!rc) rc = gipuma_hip_get_state(s, norm4_out, cost_out);
    if (timing) *timing = t;
    std::string keep = g_err;
    gipuma_hip_destroy(s);
    g_err = keep;
    return rc;
}

}  // extern "C"

#ifdef GIPUMA_HIP_FLAVOUR_TU
#pragma GCC visibility pop
#endif
