// pm_core.h -- the problem block, the numerical model M1-M4 (RNG, exp, reciprocal, bilinear taps),
// plane / ray geometry, the literal homography, the LDS layout constants and experiment switches, and the
// window-packed source views (pack kernels).  Part of the device code of the PatchMatch path (pm_device.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// PM_APPROX = 1 builds the TOLERANCE-JUDGED flavour of every kernel (GIPUMA_HIP_FLAG_FAST, gipuma_hip_fast.hip): the
// same algorithm, schedule, memory layout and random numbers, with the numerical model of rounds 1-5 (pm_sample.h,
// PM_MODEL 0: x * (1/z) for x / z, fused multiply-adds in the sample loop) and arithmetic shortcuts of the kind the
// reference takes by being built with --use_fast_math (CMakeLists.txt:23), measured per ingredient on an MI355X
// (profiles/r05_fast_mode_ingredients.txt):
//   * v_rcp_f32 without the correcting Newton step for 1/z of the warped point (1 ulp instead of correctly rounded);
//   * no proof that the window's denominators are in the range where that reciprocal is exact -- one instantiation of
//     every sample loop instead of two;
//   * the nine divisions by the plane offset in getHomography_cu as one reciprocal + a Markstein correction each.
// Measured and rejected (compile options for A/B builds only): -DPM_APPROX_HFOLD, the homography from host-folded per-view
// products (22 instead of ~200 instructions, but its rounding moves a whole window coherently by ~1e-4 px: 0.5 % of
// config C's and 7.6 % of config B's pixels leave the tolerance); -DPM_APPROX_TREE_SUM, butterfly sums instead of the
// column-per-lane kernels' relay (no time gained, 4 % of a 320x256 frame lost: the early half-sweeps decide trajectories).
// Its results are NOT bit-identical to the default flavour's; they are judged by the fraction of pixels inside 1e-4 / 1e-3
// (tests/test_fast_mode.py, tests/test_headline_parity.py).  PM_APPROX = 0 (default) is the exact flavour.
#ifndef PM_APPROX
#define PM_APPROX 0
#endif
// PM_LITERAL = 1 builds the REFERENCE-ORDER flavour (GIPUMA_HIP_FLAG_LITERAL, gipuma_hip_literal.hip): pm_sample.h's
// PM_MODEL 7 -- every tap its own bilinear fetch at the coordinates gipuma.cu:251-253 writes, x / z and y / z correctly
// rounded (config.h:44-47), H*(x, y, 1), dis and the cost accumulation as unfused multiply-adds (config.h:150-162,
// gipuma.cu:272-274, 672).  Everything else already is the reference's order.  Its results equal the reference's OWN code
// (compiled for the CPU with fp32 filter weights: the test infrastructure's _ref build) in every bit, through the same
// kernels and schedule as the default flavour (PM_MODEL 6, which differs in the taps only).
#ifndef PM_LITERAL
#define PM_LITERAL 0
#endif

namespace pm {


constexpr float kMaxCost = 1000.0f;  // config.h:22
constexpr int kMaxViews = 32;        // gipuma.cu:736
constexpr int kThreads = 256;        // 4 wavefronts per workgroup
constexpr int kTileW = 32;           // pixels per tile row (both colours)
constexpr int kSweepTileH = 16;      // 32x16 tile, 256 pixels of one colour
constexpr int kDenseTileH = 8;       // 32x8 tile, all 256 pixels (init / eval / finalize)

// A pointer into device global memory that lives in the problem block.  Loaded from memory, a plain pointer is a GENERIC
// one to the compiler: every access becomes a flat_load / flat_store, which counts on lgkmcnt as well as vmcnt and -- a
// store or atomic is never waited for -- turns every later LDS wait of the kernel into lgkmcnt(0) (round 4: one counter
// atomic anywhere in the box-25 fused kernel serialised the 338 LDS reads of every chain, config D 17 % slower).  On the
// device the wrapper converts to an address_space(1) pointer, so the accesses are global_load / global_store; on the
// host it is the plain pointer.  Same size and layout either way.
template <class T>
struct DevPtr {
    T *raw;
    using G = __attribute__((address_space(1))) T *;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PM_DEVPTR_GENERIC)  // (-DPM_DEVPTR_GENERIC: the flat accesses, for A/B runs)
    __device__ __forceinline__ operator G() const { return (G)raw; }
#else
    __host__ __device__ operator T *() const { return raw; }
#endif
    __host__ __device__ DevPtr &operator=(T *q)
    {
        raw = q;
        return *this;
    }
};

struct RefCam {  // Camera_cu of view 0, camera.h:7-62
    float K_inv[9], M_inv[9], R_orig_inv[9];
    float P_col34[3], C[3];
    float fx, cx, cy, alpha, f, baseline, depth_min, depth_max;
};
struct ViewCam {  // Camera_cu of a selected source view + its image plane
    float K[9], R[9], t[3];
    int pad;
#if PM_APPROX && defined(PM_APPROX_HFOLD)
    float A[9], u[3];  // K R K_ref^-1 and K t, formed once per view on the host in double (homography(), approx flavour)
#endif
    DevPtr<const float> img;        // float plane (row-major, Problem::pitch)
    DevPtr<const uint32_t> packed;  // window-packed u8 copy (see pack_kernel), or nullptr
};
struct Problem {  // lives in device memory, read through scalar loads (wave-uniform)
    int rows, cols, pitch, n_sel;
    int box_h, box_v, n_best, cost_comb;
    float alpha, tau_color, tau_gradient, gamma;
    float min_disp, max_disp, good_factor;
    uint32_t seed;
    DevPtr<const float> ref;
    int pw, channels;  // packed layout: texels per row of V (cols + 8); 1 = gray, 4 = colour
    int magic_addr;    // gray packed planes small enough (< 2^21 words) for float-encoded offsets
    DevPtr<unsigned char> changed;  // per pixel: did its plane change in its colour's last half-sweep (history rule)
    // early termination of refinement evaluations (see multiview_cost): enabled by the host when every
    // view cost is provably finite and below MAXCOST; theta of refinement step 0, 1, 2+
    int et_enable;
    float et_theta[3];
    // per (tile, wavefront, refinement step): > 0 while bounding the evaluation recently did not pay
    // there (a wavefront had to redo lanes); performance only, any content gives the same results
    DevPtr<unsigned char> et_hint;
    // [3 rotating slots][kEtSlot words]: what the probe workgroups (every 16th) measured per bounded
    // refinement step -- window columns a full evaluation takes, columns evaluated with the bound incl.
    // redos, (candidate, view) items left after phase 1 of refine_two_phase, items, phase-1 length used;
    // the other workgroups bound a step only if that paid for the previous half-sweep's probes.
    // Half-sweep k (= phase) writes slot k % 3, reads slot (k-1) % 3 and clears slot (k+1) % 3.
    // Performance only: any content gives the same results.
    DevPtr<unsigned> et_stat;
    int tp_g0;  // window columns of phase 1 of refine_two_phase (0: default, 3/8 of the window)
    // [8][rows*cols]: cost of neighbour slot k's plane at the pixel, left by pm::push_kernel (pm_push.h)
    // after the previous half-sweep for the pixels of the other colour; read instead of evaluated
    // when the host sets Tune::kPushConsume
    DevPtr<float> push_cost;
    // lower-bound prefilter of refinement candidates (lb_item): per pixel the kLbMax window samples with the
    // largest support weights, two bytes each (window column, window row), as kLbDwords planes of
    // rows*cols words (weight_order_kernel); lb_k > 0: samples to use (even), 0: chosen from the probes'
    // statistics, < 0: prefilter off.  Performance only: ANY list gives the same results.
    DevPtr<const uint32_t> worder;
    int lb_k;
    // rule (S) of the sweep kernels' exact skipping (colour sessions only, see gipuma_hip_create): per pixel a ring
    // of the last kSeenRing planes its propagation evaluated ([kSeenRing][rows*cols] float4) and one byte of ring
    // state (next slot | 8 once full); nullptr: rule off.  Cleared by the host whenever planes are (re-)installed.
    DevPtr<float4> seen_ring;
    DevPtr<unsigned char> seen_pos;
    // experiment aid (GIPUMA_HIP_COUNTS=1): [64 phases][kDbgSlots] event counters, or nullptr
    DevPtr<unsigned long long> dbg;
    // dispatch order of the fused launches (performance only: ANY permutation gives the same results; nullptr: off).
    // tile_clock[colour][tile][2] = 100 MHz clock at the start of the tile's workgroup / at the end of its last wavefront in
    // the colour's latest fused launch; tile_order[b] = the workgroup id (of the plain mapping, same XCD) that workgroup b
    // stands in for, written by pm::tile_order_kernel before a fused launch from the same colour's previous durations
    DevPtr<unsigned long long> tile_clock;
    DevPtr<int> tile_order;
#ifdef PM_CHECKED  // (the bounds-checked test build, below: out-of-bounds accesses per class of access)
    DevPtr<unsigned long long> viol;
#endif
#ifdef PM_WG_TICKS  // (experiment build: per workgroup of a fused launch {start, last wavefront's end, first wavefront's end}, 100 MHz)
    DevPtr<unsigned long long> wg_ticks;
#endif
    RefCam rc;
    ViewCam view[kMaxViews];
};

// ---------------------------------------------------------------------------------------------
// M4: counter-based uniform in (0,1] (stands in for curand_uniform, gipuma.cu:138-141)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x7feb352dU;
    h ^= h >> 15;
    h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}
// the (seed, phase, y, x) prefix is hashed once per pixel, the draw index per number
__device__ __forceinline__ uint32_t rng_prefix(uint32_t seed, uint32_t phase, uint32_t x, uint32_t y)
{
    uint32_t h = mix32(seed + 0x9E3779B9U);
    h = mix32(h ^ (phase + 0x85EBCA6BU));
    h = mix32(h ^ (y + 0xC2B2AE35U));
    h = mix32(h ^ (x + 0x27D4EB2FU));
    return h;
}
__device__ __forceinline__ float rng_uniform(uint32_t prefix, uint32_t draw)
{
    const uint32_t h = mix32(prefix ^ (draw + 0x165667B1U));
    return (float)((h >> 8) + 1U) * 5.9604644775390625e-8f;
}
__device__ __forceinline__ float between(float u, float lo, float hi) { return u * (hi - lo) + lo; }

// ---------------------------------------------------------------------------------------------
// M2: exp of the adaptive support weight (weight_cu, gipuma.cu:186-193)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float exp_model(float x)
{
    if (!(x >= -86.0f)) return 0.0f;
    if (x > 86.0f) x = 86.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float e = __builtin_fmaf(p, r * r, r) + 1.0f;
    return __int_as_float(__float_as_int(e) + (((int)n) << 23));
}

// ---------------------------------------------------------------------------------------------
// planes and rays (reference-camera frame)
// ---------------------------------------------------------------------------------------------
struct Vec3 {
    float x, y, z;
};

// matvecmul4, config.h:163-176
__device__ __forceinline__ Vec3 matvec(const float *m, Vec3 v)
{
    Vec3 o;
    o.x = m[0] * v.x + m[1] * v.y + m[2] * v.z;
    o.y = m[3] * v.x + m[4] * v.y + m[5] * v.z;
    o.z = m[6] * v.x + m[7] * v.y + m[8] * v.z;
    return o;
}
__device__ __forceinline__ float dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// normalize_cu, gipuma.cu:113-120 (rsqrtf -> 1/sqrtf, M2)
__device__ __forceinline__ Vec3 normalize3(Vec3 v)
{
    const float ns = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv = 1.0f / __builtin_sqrtf(ns);
    v.x *= inv;
    v.y *= inv;
    v.z *= inv;
    return v;
}
// getViewVector_cu, gipuma.cu:80-89, 122-130
__device__ __forceinline__ Vec3 view_vector(const RefCam &rc, int x, int y)
{
    Vec3 pt;
    pt.x = (float)x - rc.P_col34[0];
    pt.y = (float)y - rc.P_col34[1];
    pt.z = 1.0f - rc.P_col34[2];
    Vec3 v = matvec(rc.M_inv, pt);
    v.x = v.x - rc.C[0];
    v.y = v.y - rc.C[1];
    v.z = v.z - rc.C[2];
    return normalize3(v);
}
// vecOnHemisphere_cu, gipuma.cu:131-137
__device__ __forceinline__ Vec3 on_hemisphere(Vec3 v, Vec3 view)
{
    if (dot3(v, view) > 0.0f) {
        v.x = -v.x;
        v.y = -v.y;
        v.z = -v.z;
    }
    return v;
}
// getD_cu, gipuma.cu:96-111
__device__ __forceinline__ float plane_d(const RefCam &rc, Vec3 n, int x, int y, float depth)
{
    Vec3 pt;
    pt.x = depth * (float)x - rc.P_col34[0];
    pt.y = depth * (float)y - rc.P_col34[1];
    pt.z = depth - rc.P_col34[2];
    const Vec3 X = matvec(rc.M_inv, pt);
    return -(dot3(n, X));
}
// getDisparity_cu / getDepthFromPlane3_cu, gipuma.cu:694-715
__device__ __forceinline__ float depth_from_plane(const RefCam &rc, float4 pl, int x, int y)
{
    const float d = pl.w;
    if (d != d) return 1000.0f;
    return -d * rc.fx /
           ((pl.x * ((float)x - rc.cx)) + (pl.y * ((float)y - rc.cy)) * rc.alpha + pl.z * rc.fx);
}
// disparityDepthConversion_cu, gipuma.cu:66-68
__device__ __forceinline__ float disp_depth(float f, float baseline, float d) { return f * baseline / d; }

// v_rcp_f32 + one Newton step: the correctly rounded 1 / z for every z with biased exponent 1..252 (see rcp_newton below,
// which is this function in the exact flavour)
__device__ __forceinline__ float rcp_correct(float z)
{
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// getHomography_cu, gipuma.cu:339-356:  H = K_to * ((R_to - t_to n^T / d) * K_ref^-1).
// K, R, t, K_inv are wave-uniform (SGPR operands); n, d are per lane.
__device__ __forceinline__ void homography(const float *Kinv_ref, const ViewCam &to, float4 pl, float *H)
{
#if PM_APPROX && defined(PM_APPROX_HFOLD)
    // (measured and NOT the default, DESIGN.md 3a: the same matrix with the per-view factors folded on the host,
    //  H = A - u m^T, A = K R K_ref^-1, u = K t, m = K_ref^-T n / d -- 22 instead of 270 instructions, and no less accurate
    //  than the literal order against the true matrix, but its rounding differs from the literal order's by ~1e-4 px
    //  COHERENTLY over a whole window, which flips 10-100x more near-ties than per-sample rounding noise does)
    const float rd = __builtin_amdgcn_rcpf(pl.w);
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
        m[c] = __builtin_fmaf(pl.z, Kinv_ref[c + 6], __builtin_fmaf(pl.y, Kinv_ref[c + 3], pl.x * Kinv_ref[c])) * rd;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) H[3 * r + c] = __builtin_fmaf(-to.u[r], m[c], to.A[3 * r + c]);
    return;
#endif
    float a[9], b[9];
    const float n[3] = {pl.x, pl.y, pl.z};
#if PM_APPROX
    // the literal order with the nine IEEE divisions by the plane offset replaced by ONE correctly rounded reciprocal and
    // a Markstein correction per quotient (q0 = x r, q = q0 + (x - d q0) r): 4 instead of ~10 instructions each, and the
    // bits of x / d wherever nothing under- or overflows (scripts/exp/div_shared_denominator.c: 4.3e9 operand pairs
    // without a difference; the exact flavour does not use it because the proof for every operand pair is owed).  The
    // matrix, and with it the position of every window, then equals the exact flavour's.
    const float rdw = rcp_correct(pl.w);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float x = to.t[r] * n[c];
            const float q0 = x * rdw;
            a[3 * r + c] = to.R[3 * r + c] - __builtin_fmaf(__builtin_fmaf(-pl.w, q0, x), rdw, q0);
        }
#else
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) a[3 * r + c] = to.R[3 * r + c] - (to.t[r] * n[c]) / pl.w;
#endif
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            b[3 * r + c] = a[3 * r] * Kinv_ref[c] + a[3 * r + 1] * Kinv_ref[c + 3] + a[3 * r + 2] * Kinv_ref[c + 6];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            H[3 * r + c] = to.K[3 * r] * b[c] + to.K[3 * r + 1] * b[c + 3] + to.K[3 * r + 2] * b[c + 6];
}

__device__ __forceinline__ float lerp(float a, float t0, float t1) { return __builtin_fmaf(a, t1 - t0, t0); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ---------------------------------------------------------------------------------------------
// LDS layout of a workgroup.  One __shared__ array; the kernels carve it:
//   [0, L)                       support-weight table: gray L = 256, w(k) = exp(-k/gamma), k = |dI|;
//                                colour L = 768, k = |dB|+|dG|+|dR|
//   [L, L + 4*tw*th)             reference tile with halo R = (box+1)/2 (gipuma.cu:1844-1855), one
//                                float4 per texel: gray {I, gx1, gy1, I}, colour {B, G, R, 0}
//   gray only: + tw*th           scratch plane used while the gradients are formed
//   [.., + n_sel*256)            per-lane view-cost columns (only for the generic combiner)
// ---------------------------------------------------------------------------------------------
constexpr int kLutSize = 256;  // gray: |dI| = 0..255
template <int CH>
__host__ __device__ constexpr int lut_size()
{
    return CH == 4 ? 768 : kLutSize;  // colour: |dB|+|dG|+|dR| = 0..765
}

// sweep kernel scratch per workgroup: [8][256] candidate costs, 2048 u16 task slots, counters
constexpr int kTaskScratchFloats = 8 * kThreads + (8 * kThreads) / 2 + 32 + 128;
// ... reused by the two-phase refinement (refine_two_phase): partial view costs of a group of
// kTpViews views, the candidates, their bounds, the list of surviving (task, view) items, counters
constexpr int kTpViews = 5;
// Problem::et_stat: per rotating slot, [4 * step + {0: columns of full evaluations, 1: columns evaluated,
// 2: items left after phase 1, 3: items}] for the three bounded steps, then [12 + step] = the phase-1
// length the probes used
constexpr unsigned kEtSlot = 24;  // ... and [16 + step] = the prefilter length (lb_item) they used
constexpr int kLbMax = 32, kLbDwords = kLbMax / 2;
constexpr int kLbRegDwords = 8;  // lists of up to 16 samples travel in registers (lb_item)
template <int BOX>
__host__ __device__ constexpr int lb_max();
template <int BOX>
__host__ __device__ constexpr bool lb_in_registers()
{
#ifdef PM_LB_FORCE_STREAM  // (A/B builds)
    return false;
#else
    return lb_max<BOX>() <= 2 * kLbRegDwords;
#endif
}
template <int BOX>
__host__ __device__ constexpr int lb_max()  // samples listed per pixel: a quarter of the window, at most kLbMax
{
    constexpr int S = ((BOX + 1) / 2) * ((BOX + 1) / 2);
    return S / 4 >= kLbMax ? kLbMax : ((S / 4) & ~1);
}
constexpr int kSeenRing = 8;
// Problem::dbg slots: propagation tasks, candidates removed by rule (S), refinement items, items the
// prefilter / phase 1 left open, refinement candidates redone, refinement candidates
constexpr int kDbgSlots = 8;
enum { kDbgTasks = 0, kDbgSeen = 1, kDbgItems = 2, kDbgItemsOpen = 3, kDbgRedo = 4, kDbgCands = 5 };
struct TpLayout {  // offsets in floats into the scratch region
    static constexpr int acc = 0;                             // [kTpViews][256] partial / final view costs
    static constexpr int plane = acc + kTpViews * kThreads;   // [256] float4 candidate planes
    static constexpr int tau = plane + 4 * kThreads;          // [256] bounds
    static constexpr int items = tau + kThreads;              // [kTpViews * 256] u16: view in group << 8 | task
    static constexpr int cnt = items + kTpViews * kThreads / 2;  // two item counters, used alternately
    static constexpr int total = cnt + 4;
};
static_assert(TpLayout::total <= kTaskScratchFloats, "two-phase refinement scratch");
static_assert(TpLayout::cnt >= 8 * kThreads + (8 * kThreads) / 2 + 32, "the item counters must not alias the task scratch");

template <int CH>
__host__ __device__ constexpr int work_floats(int tile_texels, bool sweep)
{
    // gray staging plane (tile_texels floats, dead after stage_tile) and the sweep kernel's
    // per-wavefront task scratch share one region
    const int plane = CH == 1 ? tile_texels : 0;
    // dense kernels: 256 planes + 256 costs exchanged by the column-per-lane evaluation, the evaluation order (256 u16)
    // and its bucket counters (disparity_order, pm_sweep.h)
    const int tasks = sweep ? kTaskScratchFloats : 6 * kThreads;
    return plane > tasks ? plane : tasks;
}

template <int BOX>
struct Win {  // window geometry: compile-time for the shipped block sizes, runtime for BOX == 0
    int bh, bv;
    __device__ __forceinline__ Win(const Problem *P) : bh(P->box_h), bv(P->box_v) {}
    __device__ __forceinline__ int hrad() const { return BOX ? (BOX - 1) / 2 : (bh - 1) / 2; }
    __device__ __forceinline__ int vrad() const { return BOX ? (BOX - 1) / 2 : (bv - 1) / 2; }
    __device__ __forceinline__ int halo_w() const { return BOX ? (BOX + 1) / 2 : (bh + 1) / 2; }
    __device__ __forceinline__ int halo_h() const { return BOX ? (BOX + 1) / 2 : (bv + 1) / 2; }
};

// inner (y) sample loop unrolling, measured on config C (box 15): 1 -> 14.6 ms (80 VGPRs),
// 2 -> 13.3 ms (102), 4 -> 13.7 ms (128), full 8 -> 13.95 ms per sweep.  Two samples in flight
// hide the window load behind the previous sample's arithmetic without costing occupancy.
template <int BOX>
__host__ __device__ constexpr int unroll_j()
{
    return BOX == 0 ? 1 : 2;  // box 25 (13 samples per column): 89.6 -> 85.3 ms per sweep of config D
}

struct Tune {  // experiment switches (GIPUMA_HIP_TUNE), all default off
    static constexpr unsigned kNoLut = 1, kNoInterior = 2, kNoXcdRemap = 4, kGenericBox = 8,
                              kGenericCombine = 16, kRowMajorTiles = 32, kNoSkip = 64,
                              kUntrustedCosts = 128,  // set by the host after gipuma_hip_set_state
                              kPushConsume = 1u << 18,   // host-internal: propagation costs come from Problem::push_cost
                              kNoDispSort = 1u << 20,    // push / column-per-lane kernels: tasks in lane order, not by disparity bucket
                              kPlainColumnOrder = 1u << 21, // tile columns of a band in place (default: the frame's last column right after the first, tile_of)
                              kNoSeen = 1u << 22,        // no skip rule (S) (planes this pixel evaluated before)
                              kNoTwoPhase = 1u << 19,    // refinement bounded per wavefront (v11) instead of two-phase (refine_two_phase)
                              kNoMagicAddr = 1u << 30,  // integer window addressing (bits 8..17: band height)
                              kOwnerMajorTasks = 1u << 29,   // always owner-major task lists
                              kSourceMajorTasks = 1u << 28,  // always source-major (default: by iteration)
                              kNoColsKernel = 1u << 27,      // never the column-per-lane kernel
                              kColsAlways = 1u << 26,        // ... or in every iteration (default: 0 and 1)
                              kAccumChanged = 1u << 31,      // host-internal: OR into Problem::changed (second launch of a split half-sweep)
                              kNoEarlyExit = 1u << 25,       // no early termination of view costs
                              kHistorySkip = 1u << 24,       // set by the host when the history rule is valid
                              kNoHistory = 1u << 23;         // never use it
};

// ---------------------------------------------------------------------------------------------
// Window-packed source views (U8 mode).
//
// gfx950 has no texture unit, and a bilinear tap set needs a 4x4 texel window (minus corners) at
// an arbitrary position per lane: 12 scattered dword gathers per sample on a plain float plane,
// which is what bounds the float path (vector-L1 address rate).  When every image is integer
// valued in [0,255] (8-bit input converted to float, reference main.cpp:941 -- the only input the
// reference has), each source view is re-laid out ONCE per session as a "vertical-quad" image:
//
//   padded image  Pd(Y, X) = I(clamp(Y-3), clamp(X-3)),   X in [0, cols+8), Y in [0, rows+6)
//   V[Y][X] (one 32-bit word) = bytes { Pd(Y+r, X) : r = 0..3 },  Y in [0, rows+3)
//
// The 4x4 window whose top-left texel is Pd(Y, X) is the four consecutive words V[Y][X..X+3]
// (word c = column X+c, byte r = row Y+r): ONE global_load_dwordx4 at a 4-byte aligned address.
// The 3-texel replicated border makes clamp-to-edge addressing implicit, so there is
// no border branch.  V takes 4 B/pixel, the size of the float plane it stands for; neighbouring
// lanes (2 px apart) share cache lines.  (float)byte is exact, so the arithmetic is
// bit-identical to the float path.
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// pointers read out of the Problem block are generic to the compiler; these casts state that the
// image planes live in global memory so that global_load (not flat_load) is emitted
typedef const __attribute__((address_space(1))) u32x4_a4 *gptr_u32x4;
typedef const __attribute__((address_space(1))) float *gptr_f32;
typedef const __attribute__((address_space(1))) char *gptr_bytes;
// Event counters (Problem::dbg) are bumped through a GLOBAL-address-space atomic: through the generic pointer the compiler
// emits flat_atomic, which counts on lgkmcnt as well as vmcnt and, never being waited for, turns every later LDS wait of
// the kernel into lgkmcnt(0) (measured in round 4: config D 17 % slower with one such atomic anywhere in the kernel).
__device__ __forceinline__ void dbg_add(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// -DPM_CHECKED (a TEST build: scripts/build_variant.sh checked -DPM_CHECKED; no GPU AddressSanitizer on this pool): every index
// or byte offset into a global buffer that the kernels COMPUTE -- the window loads of the packed views, the state planes,
// the pushed costs, the change flags and rings -- is compared with the extent of its buffer before the access.  A
// violation is counted per class of access in Problem::viol (the host reports the counts when the session is destroyed:
// checked_collect in gipuma_hip.hip) and the access goes to element 0 instead.  Without the macro PM_AT(...) is its index and nothing else changes.
enum CheckSite { kChkWindow = 0, kChkWindowInt = 1, kChkWindowC4 = 2, kChkNorm4 = 3, kChkCost = 4, kChkPushCost = 5, kChkFlags = 6 };
#ifdef PM_CHECKED
template <typename I>
__device__ __forceinline__ I pm_chk(const Problem *P, I idx, size_t n, int site)
{
    if ((size_t)idx < n) return idx;
    dbg_add(&P->viol[site], 1ull);
    return (I)0;
}
#define PM_AT(P, idx, n, site) pm::pm_chk(P, idx, (size_t)(n), site)
#else
#define PM_AT(P, idx, n, site) (idx)
#endif
#define PM_NP(P) ((size_t)(P)->rows * (size_t)(P)->cols)

// byte -> float.  Spelled as the hardware instruction so that the compiler keeps ONE half-rate
// conversion per texel (it otherwise rewrites (float)b1 - (float)b0 into a byte-select
// subtraction plus a conversion: two half-rate ops for what one full-rate v_sub_f32 does).
#define PM_UB(n)                                                                \
    __device__ __forceinline__ float ub##n(uint32_t w)                          \
    {                                                                           \
        float f;                                                                \
        asm("v_cvt_f32_ubyte" #n " %0, %1" : "=v"(f) : "v"(w));                 \
        return f;                                                               \
    }
PM_UB(0) PM_UB(1) PM_UB(2) PM_UB(3)
#undef PM_UB
// float -> uint32 with saturation (negative and NaN -> 0, huge -> 0xffffffff): v_cvt_u32_f32.
// One instruction replaces fmax + fmin + cvt of the clamped conversion.
__device__ __forceinline__ uint32_t cvt_u32_sat(float x)
{
    uint32_t u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(x));
    return u;
}

// grid = (ceil(pw/256), rows+3)
__global__ __launch_bounds__(kThreads) void pack_kernel(const float *__restrict__ img, int rows, int cols,
                                                        int pitch, int pw, uint32_t *__restrict__ packed)
{
    const int X = blockIdx.x * kThreads + threadIdx.x;
    const int Y = blockIdx.y;
    if (X >= pw) return;
    const int x = clampi(X - 3, 0, cols - 1);
    uint32_t w = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = clampi(Y + r - 3, 0, rows - 1);
        const float v = img[y * pitch + x];
        w |= ((uint32_t)v & 0xffu) << (8 * r);
    }
    packed[(size_t)Y * pw + X] = w;
}

// flag[0] |= 1 if any value of the plane is not an integer in [0,255]
__global__ __launch_bounds__(kThreads) void check_u8_kernel(const float *__restrict__ img, int rows, int cols,
                                                            int pitch, int *__restrict__ flag)
{
    const int x = blockIdx.x * kThreads + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const float v = img[y * pitch + x];
    if (!(v >= 0.0f && v <= 255.0f) || v != __builtin_floorf(v)) atomicOr(flag, 1);
}

// 1/z of the warped point (M2: an IEEE-correct reciprocal).  v_rcp_f32 followed by one Newton step
// is bit-identical to the correctly rounded 1.0f/z for EVERY input whose biased exponent is
// 1..252, i.e. 2^-126 <= |z| < 2^126 (exhaustive check over all 2^32 inputs:
// scripts/ubench/rcp_exact.hip and gipuma_hip_selftest_reciprocal(), run by the gpu tests).  It is
// 3 instructions against the 10 of the div_scale/div_fmas/div_fixup expansion.  view_cost uses it
// only after proving the whole window stays inside a (much narrower) safe range.
__device__ __forceinline__ float rcp_newton(float z)
{
#if PM_APPROX && !defined(PM_APPROX_KEEP_NEWTON)  // (-DPM_APPROX_KEEP_NEWTON: A/B builds, +0.5 % time)
    return __builtin_amdgcn_rcpf(z);  // 1 ulp; no correction step
#endif
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
template <bool FAST>
__device__ __forceinline__ float recip(float z)
{
#if PM_APPROX
    return rcp_newton(z);
#endif
    return FAST ? rcp_newton(z) : 1.0f / z;
}
// counts inputs in [lo_exp, hi_exp] (biased exponents) where rcp_newton != 1.0f/z
__global__ __launch_bounds__(kThreads) void rcp_selftest_kernel(unsigned long long *bad, uint32_t lo_exp,
                                                                uint32_t hi_exp)
{
    const uint32_t hi = blockIdx.x;
    unsigned c = 0;
    for (uint32_t lo = threadIdx.x; lo < 65536; lo += kThreads) {
        const uint32_t bits = (hi << 16) | lo;
        const uint32_t ex = (bits >> 23) & 0xffu;
        const float z = __uint_as_float(bits);
        if (ex >= lo_exp && ex <= hi_exp) c += __float_as_uint(rcp_newton(z)) != __float_as_uint(1.0f / z);
    }
    if (c) atomicAdd(bad, (unsigned long long)c);
}

// fminf(x, tau) as the bare v_min_f32.  The compiler's fminf first quiets a possible signalling NaN
// in `tau` with a v_max(tau, tau) that it re-issues inside the sample loop; here x is always an
// arithmetic result (never signalling) and for a quiet-NaN x v_min_f32 returns tau like fminf does.
__device__ __forceinline__ float min_nc(float x, float tau)
{
    float r;
    // tau is wave-uniform (a kernel parameter): taken as a scalar operand, no v_mov per use
    asm("v_min_f32_e64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(tau));
    return r;
}
__device__ __forceinline__ float min_abs_nc(float x, float tau)  // fminf(fabsf(x), tau)
{
    float r;
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(tau));
    return r;
}

// The five bilinear taps of pmCostComputation_shared (gipuma.cu:251-253) from one 4x4 window
// (M1): centre value and the +-1 texel differences in x and y.  t<row><col>, corners unused.
struct Taps {
    float sc, gx2, gy2;
};
__device__ __forceinline__ Taps taps12(float a, float b, float t01, float t02, float t10, float t11, float t12,
                                       float t13, float t20, float t21, float t22, float t23, float t31,
                                       float t32)
{
    // Columns interpolated along y first, the +-1 differences taken on the texels BEFORE the interpolation (bilinear
    // interpolation is linear in the texels): 24 operations where five separate taps take 28.  This order IS the
    // numerical model's (M1; the CPU restatement computes the same expressions).
    const float V0 = lerp(b, t10, t20), V1 = lerp(b, t11, t21), V2 = lerp(b, t12, t22), V3 = lerp(b, t13, t23);
    const float W1 = lerp(b, t21 - t01, t31 - t11), W2 = lerp(b, t22 - t02, t32 - t12);
    Taps o;
    o.sc = lerp(a, V1, V2);
    o.gx2 = lerp(a, V2 - V0, V3 - V1);
    o.gy2 = lerp(a, W1, W2);
    return o;
}
// window words w0..w3 = columns X..X+3, byte r = row Y+r
__device__ __forceinline__ Taps taps_u8(float a, float b, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    return taps12(a, b, ub0(w1), ub0(w2), ub1(w0), ub1(w1), ub1(w2), ub1(w3), ub2(w0), ub2(w1), ub2(w2),
                  ub2(w3), ub3(w1), ub3(w2));
}

#if PM_LITERAL
// tex2D<float>(tex, x, y) with cudaFilterModeLinear on unnormalised coordinates and clamp addressing (main.cpp:641-648),
// fp32 filter weights: xB = x - 0.5, i = floor(xB), a = xB - i, the four texels blended by fmaf lerps -- the texture model
// of the reference-on-CPU build the tests compare with (its ref_tex_channel) and of the CPU restatement's literal flavour
// (STRIDE floats per texel: 1 for gray planes, 4 for float4 texels -- tex2D<float4> filters every component alike)
template <int STRIDE = 1>
__device__ __forceinline__ float tex2d_literal(gptr_f32 img, int rows, int cols, int pitch, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb);
    const float a = xb - fx, b = yb - fy;
    const int ix = (int)__builtin_fminf(__builtin_fmaxf(fx, -2.0f), (float)cols);
    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fy, -2.0f), (float)rows);
    const int c0 = STRIDE * clampi(ix, 0, cols - 1), c1 = STRIDE * clampi(ix + 1, 0, cols - 1);
    const int r0 = clampi(iy, 0, rows - 1) * pitch, r1 = clampi(iy + 1, 0, rows - 1) * pitch;
    const float t00 = img[r0 + c0], t10 = img[r0 + c1], t01 = img[r1 + c0], t11 = img[r1 + c1];
    const float v0 = __builtin_fmaf(a, t10 - t00, t00), v1 = __builtin_fmaf(a, t11 - t01, t01);
    return __builtin_fmaf(b, v1 - v0, v0);
}
#endif

// colour (float4 texels: B, G, R, unused): V3[Y][X][c], c = 0..2 -- the window of all three
// channels is 12 consecutive words (word 3k+c = column X+k, channel c): three dwordx4 loads
__global__ __launch_bounds__(kThreads) void pack_kernel_c4(const float *__restrict__ img, int rows, int cols,
                                                           int pitch, int pw, uint32_t *__restrict__ packed)
{
    const int X = blockIdx.x * kThreads + threadIdx.x;
    const int Y = blockIdx.y;
    if (X >= pw) return;
    const int x = clampi(X - 3, 0, cols - 1);
    uint32_t w[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = clampi(Y + r - 3, 0, rows - 1);
#pragma unroll
        for (int c = 0; c < 3; c++) w[c] |= ((uint32_t)img[y * pitch + 4 * x + c] & 0xffu) << (8 * r);
    }
    uint32_t *o = packed + ((size_t)Y * pw + X) * 3;
    o[0] = w[0];
    o[1] = w[1];
    o[2] = w[2];
}

__global__ __launch_bounds__(kThreads) void check_u8_kernel_c4(const float *__restrict__ img, int rows, int cols,
                                                               int pitch, int *__restrict__ flag)
{
    const int x = blockIdx.x * kThreads + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = img[y * pitch + 4 * x + c];
        bad |= !(v >= 0.0f && v <= 255.0f) || v != __builtin_floorf(v);
    }
    if (bad) atomicOr(flag, 1);
}

}  // namespace pm
