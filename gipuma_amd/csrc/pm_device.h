// pm_device.h -- device code of the red-black PatchMatch path for gfx950 (MI355X).
//
// Written for CDNA4 directly: 64-wide wavefronts, one lane per pixel of the active checkerboard
// colour, the reference-image tile staged in LDS, camera constants wave-uniform (scalar loads),
// source views read with software bilinear filtering (gfx950 has no image instructions,
// SURVEY.md F1).  No MFMA: the patch cost is a small stencil reduction (BASELINE.json).
//
// The arithmetic follows the numerical model M1-M4 of DESIGN.md section 3 (fp32 lerp bilinear
// taps, exp_model, x*(1/z), explicit fmaf); build with -ffp-contract=off so only the fmaf()
// written here fuse.  The reference functions each piece stands for are cited by file:line of
// reference gipuma.cu.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

constexpr float kMaxCost = 1000.0f;  // config.h:22
constexpr int kMaxViews = 32;        // gipuma.cu:736
constexpr int kThreads = 256;        // 4 wavefronts per workgroup
constexpr int kTileW = 32;           // pixels per tile row (both colours)
constexpr int kSweepTileH = 16;      // 32x16 tile, 256 pixels of one colour
constexpr int kDenseTileH = 8;       // 32x8 tile, all 256 pixels (init / eval / finalize)

struct RefCam {  // Camera_cu of view 0, camera.h:7-62
    float K_inv[9], M_inv[9], R_orig_inv[9];
    float P_col34[3], C[3];
    float fx, cx, cy, alpha, f, baseline, depth_min, depth_max;
};
struct ViewCam {  // Camera_cu of a selected source view + its image plane
    float K[9], R[9], t[3];
    int pad;
    const float *img;        // float plane (row-major, Problem::pitch)
    const uint32_t *packed;  // window-packed u8 copy (see pack_kernel), or nullptr
};
struct Problem {  // lives in device memory, read through scalar loads (wave-uniform)
    int rows, cols, pitch, n_sel;
    int box_h, box_v, n_best, cost_comb;
    float alpha, tau_color, tau_gradient, gamma;
    float min_disp, max_disp, good_factor;
    uint32_t seed;
    const float *ref;
    int pw, channels;  // packed layout: texels per row of V (cols + 8); 1 = gray, 4 = colour
    int magic_addr;    // gray packed planes small enough (< 2^21 words) for float-encoded offsets
    unsigned char *changed;  // per pixel: did its plane change in its colour's last half-sweep (history rule)
    // early termination of refinement evaluations (see multiview_cost): enabled by the host when every
    // view cost is provably finite and below MAXCOST; theta of refinement step 0, 1, 2+
    int et_enable;
    float et_theta[3];
    // per (tile, wavefront, refinement step): > 0 while bounding the evaluation recently did not pay
    // there (a wavefront had to redo lanes); performance only, any content gives the same results
    unsigned char *et_hint;
    // [3 rotating slots][kEtSlot words]: what the probe workgroups (every 16th) measured per bounded
    // refinement step -- window columns a full evaluation takes, columns evaluated with the bound incl.
    // redos, (candidate, view) items left after phase 1 of refine_two_phase, items, phase-1 length used;
    // the other workgroups bound a step only if that paid for the previous half-sweep's probes.
    // Half-sweep k (= phase) writes slot k % 3, reads slot (k-1) % 3 and clears slot (k+1) % 3.
    // Performance only: any content gives the same results.
    unsigned *et_stat;
    int tp_g0;  // window columns of phase 1 of refine_two_phase (0: default, 3/8 of the window)
    // [8][rows*cols]: cost of neighbour slot k's plane at the pixel, left by pm::push_kernel (pm_push.h)
    // after the previous half-sweep for the pixels of the other colour; read instead of evaluated
    // when the host sets Tune::kPushConsume
    float *push_cost;
    // lower-bound prefilter of refinement candidates (lb_item): per pixel the kLbMax window samples with the
    // largest support weights, two bytes each (window column, window row), as kLbDwords planes of
    // rows*cols words (weight_order_kernel); lb_k > 0: samples to use (even), 0: chosen from the probes'
    // statistics, < 0: prefilter off.  Performance only: ANY list gives the same results.
    const uint32_t *worder;
    int lb_k;
    // rule (S) of the sweep kernels' exact skipping: per pixel a ring of the last kSeenRing planes its
    // propagation evaluated ([kSeenRing][rows*cols] float4) and one byte of ring state (next slot | 8 once
    // full); nullptr: rule off.  Cleared by the host whenever planes are (re-)installed.
    float4 *seen_ring;
    unsigned char *seen_pos;
    // experiment aid (GIPUMA_HIP_COUNTS=1): [64 phases][kDbgSlots] event counters, or nullptr
    unsigned long long *dbg;
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

// getHomography_cu, gipuma.cu:339-356:  H = K_to * ((R_to - t_to n^T / d) * K_ref^-1).
// K, R, t, K_inv are wave-uniform (SGPR operands); n, d are per lane.
__device__ __forceinline__ void homography(const float *Kinv_ref, const ViewCam &to, float4 pl, float *H)
{
    float a[9], b[9];
    const float n[3] = {pl.x, pl.y, pl.z};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) a[3 * r + c] = to.R[3 * r + c] - (to.t[r] * n[c]) / pl.w;
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
    // dense kernels: 256 planes + 256 costs exchanged by the column-per-lane evaluation
    const int tasks = sweep ? kTaskScratchFloats : 5 * kThreads;
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
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
template <bool FAST>
__device__ __forceinline__ float recip(float z)
{
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
    const float C0 = lerp(a, t01, t02);
    const float L1 = lerp(a, t10, t11), C1 = lerp(a, t11, t12), R1 = lerp(a, t12, t13);
    const float L2 = lerp(a, t20, t21), C2 = lerp(a, t21, t22), R2 = lerp(a, t22, t23);
    const float C3 = lerp(a, t31, t32);
    Taps o;
    o.sc = lerp(b, C1, C2);
    o.gx2 = lerp(b, R1, R2) - lerp(b, L1, L2);
    o.gy2 = lerp(b, C2, C3) - lerp(b, C0, C1);
    return o;
}
// window words w0..w3 = columns X..X+3, byte r = row Y+r
__device__ __forceinline__ Taps taps_u8(float a, float b, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    return taps12(a, b, ub0(w1), ub0(w2), ub1(w0), ub1(w1), ub1(w2), ub1(w3), ub2(w0), ub2(w1), ub2(w2),
                  ub2(w3), ub3(w1), ub3(w2));
}

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

// Patch cost of one source view: pmCost_shared + pmCostComputation_shared,
// gipuma.cu:585-680 and :223-277.  `tp0` points at the pixel's own texel inside the LDS tile.
// Is every warped-point denominator of the window safely inside the range where rcp_newton is
// exact?  Z(i,j) = fmaf(H7, qy, fmaf(H6, qx, H8)) is monotone in qx and in qy (one rounding each),
// so all window values lie between the four corner values.
__device__ __forceinline__ bool window_z_safe(const float *H, float qx0, float qx1, float qy0, float qy1)
{
    const float z00 = __builtin_fmaf(H[7], qy0, __builtin_fmaf(H[6], qx0, H[8]));
    const float z01 = __builtin_fmaf(H[7], qy1, __builtin_fmaf(H[6], qx0, H[8]));
    const float z10 = __builtin_fmaf(H[7], qy0, __builtin_fmaf(H[6], qx1, H[8]));
    const float z11 = __builtin_fmaf(H[7], qy1, __builtin_fmaf(H[6], qx1, H[8]));
    const float lo = __builtin_fminf(__builtin_fminf(z00, z01), __builtin_fminf(z10, z11));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(z00, z01), __builtin_fmaxf(z10, z11));
    // same sign, and magnitudes in [2^-100, 2^100] (NaN fails every comparison)
    return (lo >= 0x1p-100f && hi <= 0x1p100f) || (hi <= -0x1p-100f && lo >= -0x1p100f);
}

// Exponent trick used by the U8 loop: for an integer n in [0, 2^21), the float 2^21 + n has ulp 1/4,
// so its bit pattern is 0x4a000000 + 4n -- a byte offset of 4-byte entry n, produced by a full-rate
// fp32 add instead of cvt + shift (conversions, integer min/max and shifts issue at ~60 % of the
// fp32 rate on gfx950, scripts/ubench/valu_rates.hip).  The constant part moves into the base.
constexpr uint32_t kMagicBits = 0x4a000000u;  // bits of 2^21
constexpr float kMagicF = 0x1p21f;
constexpr int kMagicMaxWords = (1 << 21) - 8;

template <int BOX, bool U8, bool INTERIOR, bool FAST, bool MAGIC>
__device__ __forceinline__ float view_cost_loop(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py,
                                                const Win<BOX> &win)
{
    const gptr_f32 img = (gptr_f32)vc.img;
    const uint32_t *__restrict__ packed = vc.packed;
    const uint32_t pw = (uint32_t)P->pw;
    const uint32_t xmax = (uint32_t)(P->cols + 2), ymax = (uint32_t)(P->rows + 2);
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const float colsf = (float)cols, rowsf = (float)rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient, gamma = P->gamma;
    const float centre = tp0[0];
    const int hr = win.hrad(), vr = win.vrad();
    // MAGIC: offset of window (Xc, Yc), Xc in [-2, cols], Yc in [-2, rows] (clamped floor
    // coordinates; entry (Yc+2)*pw + Xc+2 of V), as the bits of fma(Yc, pw, Xc + magic_c)
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)packed - (uintptr_t)kMagicBits);
    const char *lut_magic = (const char *)lut - kMagicBits;
    float cost = 0.0f;
    // (float)(px + i) == (float)px + (float)i exactly (small integers): full-rate adds, no cvt
    float qx = (float)(px - hr);
    for (int i = -hr; i <= hr; i += 2, qx += 2.0f) {
        const float X0 = __builtin_fmaf(H[0], qx, H[2]);
        const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
        const float Z0 = __builtin_fmaf(H[6], qx, H[8]);
        float qy = (float)(py - vr);
#pragma unroll unroll_j<BOX>()
        for (int j = -vr; j <= vr; j += 2, qy += 2.0f) {
            // one ds_read_b128: {I(q), gx1(q), gy1(q)} of the reference tile
            const float4 t4 = *reinterpret_cast<const float4 *>(tp0 + 4 * (j * tw + i));
            // weight_cu, gipuma.cu:186-193
            const float leftValue = t4.x;
            const float colorDis = __builtin_fabsf(leftValue - centre);
            float w;
            if (U8)  // images are integer valued in [0,255]: 256 possible weights
                w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
            else
                w = exp_model(-colorDis / gamma);
            // getCorrespondingPoint_cu, gipuma.cu:207-217
            const float X = __builtin_fmaf(H[1], qy, X0);
            const float Y = __builtin_fmaf(H[4], qy, Y0);
            const float Z = __builtin_fmaf(H[7], qy, Z0);
            const float rz = recip<FAST>(Z);
            const float sx = X * rz, sy = Y * rz;
            // M1: five bilinear taps sharing one 4x4 texel window (gipuma.cu:251-253)
            const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
            const float a = sx - fx0, b = sy - fy0;
            Taps tp5;
            if (U8 && MAGIC) {  // U8 mode: the whole window is one 16-byte load
                // v_med3_f32 returns min3 when an input is NaN: NaN -> -2, like the saturating cvt
                const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
                const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
                const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
                const u32x4_a4 wv = *(gptr_u32x4)(magic_base + off);
                tp5 = taps_u8(a, b, wv.x, wv.y, wv.z, wv.w);
            } else if (U8) {
                // X = clamp(floor(sx), -2, cols) + 2, same for Y: the +2 is exact wherever the
                // clamp does not saturate
                const uint32_t X = min(cvt_u32_sat(fx0 + 2.0f), xmax);
                const uint32_t Y = min(cvt_u32_sat(fy0 + 2.0f), ymax);
                const uint32_t off = (Y * pw + X) << 2;
                const u32x4_a4 wv = *(gptr_u32x4)((gptr_bytes)packed + off);
                tp5 = taps_u8(a, b, wv.x, wv.y, wv.z, wv.w);
            } else {
                // float planes: keep the float->int conversion defined for huge / NaN coordinates
                const int ix = (int)__builtin_fminf(__builtin_fmaxf(fx0, -2.0f), colsf);
                const int iy = (int)__builtin_fminf(__builtin_fmaxf(fy0, -2.0f), rowsf);
                const bool inside = ix >= 1 && ix <= cols - 3 && iy >= 1 && iy <= rows - 3;
                if (INTERIOR && __all(inside)) {
                    const gptr_f32 s = img + (iy * pitch + ix);
                    tp5 = taps12(a, b, s[-pitch], s[-pitch + 1], s[-1], s[0], s[1], s[2], s[pitch - 1], s[pitch],
                                 s[pitch + 1], s[pitch + 2], s[2 * pitch], s[2 * pitch + 1]);
                } else {
                    const int c0 = clampi(ix - 1, 0, cols - 1), c1 = clampi(ix, 0, cols - 1);
                    const int c2 = clampi(ix + 1, 0, cols - 1), c3 = clampi(ix + 2, 0, cols - 1);
                    const int r0 = clampi(iy - 1, 0, rows - 1) * pitch, r1 = clampi(iy, 0, rows - 1) * pitch;
                    const int r2 = clampi(iy + 1, 0, rows - 1) * pitch, r3 = clampi(iy + 2, 0, rows - 1) * pitch;
                    tp5 = taps12(a, b, img[r0 + c1], img[r0 + c2], img[r1 + c0], img[r1 + c1], img[r1 + c2],
                                 img[r1 + c3], img[r2 + c0], img[r2 + c1], img[r2 + c2], img[r2 + c3],
                                 img[r3 + c1], img[r3 + c2]);
                }
            }
            const float sc = tp5.sc, gx2 = tp5.gx2, gy2 = tp5.gy2;
            // pmCostComputation_shared, gipuma.cu:251-274
            const float colDiff = t4.w - sc;  // t4.w == t4.x == I(q); |.| taken in the min below
            const float gx1 = t4.y;
            const float gy1 = t4.z;
            const float gradX = gx1 - gx2;
            const float gradY = gy1 - gy2;
            const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
            const float colDis = min_abs_nc(colDiff, tau_color);
            const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
            cost = __builtin_fmaf(w, dis, cost);
        }
    }
    return cost;
}

// The same loop for the shipped case (square compile-time box, gray U8 planes with float-encoded
// offsets), software-pipelined by hand: the window of sample s+2 is requested before sample s is
// reduced, across column boundaries, so that each wavefront keeps two 16-byte loads in flight
// instead of waiting for the one it has just issued.  Per sample the arithmetic and the order of
// the cost accumulation are those of view_cost_loop -- the results are bit-identical.  (The two
// requests past the last sample fetch clamped, valid addresses and are dropped.)
struct WinReq {
    float a, b;
    u32x4_a4 w;
};
template <int BOX, bool FAST, bool ET>
__device__ __forceinline__ float view_cost_pipe(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py, float tau,
                                                int *cols_done = nullptr)
{
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2, N = R + 1;  // offsets -R, -R+2, ..., R
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float centre = tp0[0];
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)vc.packed - (uintptr_t)kMagicBits);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const float H1 = H[1], H4 = H[4], H7 = H[7];

    auto request = [&](float X0, float Y0, float Z0, float qy) -> WinReq {
        // getCorrespondingPoint_cu, gipuma.cu:207-217
        const float X = __builtin_fmaf(H1, qy, X0);
        const float Y = __builtin_fmaf(H4, qy, Y0);
        const float Z = __builtin_fmaf(H7, qy, Z0);
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        WinReq r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
        const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
        r.w = *(gptr_u32x4)(magic_base + off);
        return r;
    };

    const float qy0 = (float)(py - R);
    float qx = (float)(px - R);
    float X0 = __builtin_fmaf(H[0], qx, H[2]);
    float Y0 = __builtin_fmaf(H[3], qx, H[5]);
    float Z0 = __builtin_fmaf(H[6], qx, H[8]);
    WinReq r0 = request(X0, Y0, Z0, qy0), r1 = request(X0, Y0, Z0, qy0 + 2.0f);
    float cost = 0.0f;
    const float *tcol = tp0 + 4 * (-R * tw - R);  // texel (-R, -R) of the window
    for (int c = 0; c < N; c++, tcol += 8) {
        const float qxn = qx + 2.0f;
        const float X0n = __builtin_fmaf(H[0], qxn, H[2]);
        const float Y0n = __builtin_fmaf(H[3], qxn, H[5]);
        const float Z0n = __builtin_fmaf(H[6], qxn, H[8]);
#pragma unroll
        for (int k = 0; k < N; k++) {
            const WinReq cur = r0;
            r0 = r1;
            // sample k+2 of this column, or the first two of the next one
            if (k + 2 < N)
                r1 = request(X0, Y0, Z0, qy0 + (float)(2 * (k + 2)));
            else
                r1 = request(X0n, Y0n, Z0n, qy0 + (float)(2 * (k + 2 - N)));
            __builtin_amdgcn_sched_barrier(0);  // keep the request ahead of this sample's reduction
            // one ds_read_b128: {I(q), gx1(q), gy1(q), I(q)} of the reference tile
            const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
            // weight_cu, gipuma.cu:186-193: 256 possible weights
            const float colorDis = __builtin_fabsf(t4.x - centre);
            const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
            const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
            // pmCostComputation_shared, gipuma.cu:251-274
            const float colDiff = t4.w - tp5.sc;
            const float gradX = t4.y - tp5.gx2;
            const float gradY = t4.z - tp5.gy2;
            const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
            const float colDis = min_abs_nc(colDiff, tau_color);
            const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
            cost = __builtin_fmaf(w, dis, cost);
            __builtin_amdgcn_sched_barrier(0);
        }
        qx = qxn;
        X0 = X0n;
        Y0 = Y0n;
        Z0 = Z0n;
        // early termination (ET): the partial sum only grows (w, dis >= 0, fmaf rounds monotonically),
        // so once every lane of the wavefront has reached its bound the rest of the view cannot
        // matter (see multiview_cost); the two windows already requested are dropped
        if (ET && __all(cost >= tau)) {
            if (cols_done) *cols_done -= N - 1 - c;  // (wave-uniform bookkeeping of the probe workgroups)
            break;
        }
    }
    if (ET && cols_done) *cols_done += N + 1;  // (+1: homography and set-up of the view, paid again by a redo)
    return cost;
}

// view_cost_pipe restricted to the window columns [c0, c1), continuing from the partial sum `cost`
// (the value view_cost_pipe holds after column c0 - 1): the same samples, the same instruction
// sequence per sample, the same accumulation order.  `magic_base` may differ per lane (the lanes
// of a wavefront may work on different source views, see refine_two_phase).  The wavefront leaves
// after the first column at which every lane has reached its `tau`; *cols_run += columns evaluated.
template <int BOX, bool FAST>
__device__ __forceinline__ float view_cost_pipe_range(const Problem *__restrict__ P, gptr_bytes magic_base,
                                                      const float *__restrict__ H, const float *__restrict__ tp0,
                                                      int tw, const float *__restrict__ lut, int px, int py, int c0,
                                                      int c1, float cost, float tau, int *cols_run)
{
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float centre = tp0[0];
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const float H1 = H[1], H4 = H[4], H7 = H[7];

    auto request = [&](float X0, float Y0, float Z0, float qy) -> WinReq {
        const float X = __builtin_fmaf(H1, qy, X0);
        const float Y = __builtin_fmaf(H4, qy, Y0);
        const float Z = __builtin_fmaf(H7, qy, Z0);
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        WinReq r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
        const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
        r.w = *(gptr_u32x4)(magic_base + off);
        return r;
    };

    const float qy0 = (float)(py - R);
    float qx = (float)(px - R + 2 * c0);  // (exact: small integers)
    float X0 = __builtin_fmaf(H[0], qx, H[2]);
    float Y0 = __builtin_fmaf(H[3], qx, H[5]);
    float Z0 = __builtin_fmaf(H[6], qx, H[8]);
    WinReq r0 = request(X0, Y0, Z0, qy0), r1 = request(X0, Y0, Z0, qy0 + 2.0f);
    const float *tcol = tp0 + 4 * (-R * tw - R) + 8 * c0;
    int c = c0;
    for (; c < c1; c++, tcol += 8) {
        const float qxn = qx + 2.0f;
        const float X0n = __builtin_fmaf(H[0], qxn, H[2]);
        const float Y0n = __builtin_fmaf(H[3], qxn, H[5]);
        const float Z0n = __builtin_fmaf(H[6], qxn, H[8]);
#pragma unroll
        for (int k = 0; k < N; k++) {
            const WinReq cur = r0;
            r0 = r1;
            if (k + 2 < N)
                r1 = request(X0, Y0, Z0, qy0 + (float)(2 * (k + 2)));
            else
                r1 = request(X0n, Y0n, Z0n, qy0 + (float)(2 * (k + 2 - N)));
            __builtin_amdgcn_sched_barrier(0);
            const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
            const float colorDis = __builtin_fabsf(t4.x - centre);
            const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
            const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
            const float colDiff = t4.w - tp5.sc;
            const float gradX = t4.y - tp5.gx2;
            const float gradY = t4.z - tp5.gy2;
            const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
            const float colDis = min_abs_nc(colDiff, tau_color);
            const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
            cost = __builtin_fmaf(w, dis, cost);
            __builtin_amdgcn_sched_barrier(0);
        }
        qx = qxn;
        X0 = X0n;
        Y0 = Y0n;
        Z0 = Z0n;
        if (__all(cost >= tau)) {
            c++;
            break;
        }
    }
    if (cols_run) *cols_run += c - c0;
    return cost;
}

// ---------------------------------------------------------------------------------------------
// Lower-bound prefilter of refinement candidates (performance only).
//
// The support weight w(p, q) = exp(-|I(q) - I(p)| / gamma) of a window sample depends on the reference
// image alone (weight_cu, gipuma.cu:186-193) -- not on the plane, the view or the iteration --, and a
// view cost is a sum of terms w * dis >= 0 (pmCost_shared, :633-676).  So the sum over ANY subset of a
// pixel's window samples is a lower bound of the view cost, and the subset that bounds best is the same
// for every evaluation at that pixel: the samples with the largest weights.  weight_order_kernel lists
// them once per solve (kLbMax per pixel); lb_item sums the first K of them for one (candidate, view)
// item, each term by the instruction sequence of view_cost_pipe (same bits per term).
//
// Rigour against rounding.  Let C be the reference's chain value (64 fmaf's in window order), T the exact
// real sum of its terms, l the value lb_item accumulates over S of the terms (|S| <= 16, any order) and
// T_S <= T their exact sum, u = 2^-24.  Every fmaf rounds a non-negative exact value to nearest, so
// C >= T (1-u)^64 - 64 * 2^-150 and l <= T_S (1+u)^16 + 16 * 2^-150 (the absolute terms cover subnormal
// partial sums).  Hence C >= l (1 - 81u) - 2^-143, and for l >= 2^-60
//     L' = l * (1 - 2^-16)   (one more rounding, 2^-16 = 256 u)
// satisfies L' <= C.  An item with L' >= thr is decided: its view cost is at least L', which is what
// the ViewCombiner gets -- "a lower bound >= thr", the case multiview_cost's proof calls an abandoned
// view.  Everything else about refine_two_phase is unchanged.
// ---------------------------------------------------------------------------------------------
constexpr float kLbShrink = 0.9999847412109375f;  // 1 - 2^-16
constexpr float kLbFloor = 0x1p-60f;

// one lane per pixel; order[d * np + pixel] = {col, row} of samples 2d and 2d+1 (bytes 0..3), heaviest first.
// The samples are listed in groups of PM_LB_GROUP horizontally adjacent ones (window columns G c .. G c + G - 1
// of one window row, ranked by their summed weights): the windows of a group lie next to each other in the
// source view, i.e. in one cache line, and a refinement candidate's window loads are bound by the vector
// L1's line fills (every lane has its own random plane), not by their count.
#ifndef PM_LB_GROUP
#define PM_LB_GROUP 1  // (measured on config C: singles 89.5, pairs 91.2, quads 93.0 ms per view)
#endif
template <int BOX, int CH = 1>
__global__ __launch_bounds__(kThreads) void weight_order_kernel(const Problem *__restrict__ P,
                                                                uint32_t *__restrict__ order)
{
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    constexpr int G = PM_LB_GROUP, NG = N / G, KG = kLbMax / G;  // group size, groups per window row, groups listed
    static_assert(G == 1 || G == 2 || G == 4, "group size");
    static_assert(NG * N >= KG, "the window has enough groups");
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const int np = rows * cols;
    const int center = blockIdx.x * kThreads + threadIdx.x;
    if (center >= np) return;
    const int py = center / cols, px = center - py * cols;
    const gptr_f32 ref = (gptr_f32)P->ref;
    // (colour: |dB| + |dG| + |dR|, the index of the colour weight table)
    auto texel_dist = [&](int x, int y) -> float {
        if (CH == 4) {
            const gptr_f32 a = ref + (y * pitch + 4 * x), c = ref + (py * pitch + 4 * px);
            return __builtin_fabsf(a[0] - c[0]) + __builtin_fabsf(a[1] - c[1]) + __builtin_fabsf(a[2] - c[2]);
        }
        return __builtin_fabsf(ref[y * pitch + x] - ref[py * pitch + px]);
    };
    uint32_t best[KG];  // ascending keys: sum of |dI| << 16 | row << 8 | first column
#pragma unroll
    for (int k = 0; k < KG; k++) best[k] = 0xffffffffu;
    for (int ri = 0; ri < N; ri++) {
        const int y = clampi(py - R + 2 * ri, 0, rows - 1);
        for (int c = 0; c < NG; c++) {
            float dsum = 0.0f;
#pragma unroll
            for (int e = 0; e < G; e++) {
                const int x = clampi(px - R + 2 * (G * c + e), 0, cols - 1);
                dsum += texel_dist(x, y);
            }
            uint32_t key = (min(cvt_u32_sat(dsum), 0xfffeu) << 16) | (uint32_t)(ri << 8) | (uint32_t)(G * c);
#pragma unroll
            for (int k = 0; k < KG; k++) {
                const uint32_t lo = min(best[k], key);
                key = max(best[k], key);
                best[k] = lo;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < kLbDwords; d++) {
        // samples 2d and 2d + 1: members (2d) % G and (2d + 1) % G of groups (2d) / G and (2d + 1) / G
        const uint32_t s0 = (best[(2 * d) / G] & 0xffffu) + (uint32_t)((2 * d) % G);
        const uint32_t s1 = (best[(2 * d + 1) / G] & 0xffffu) + (uint32_t)((2 * d + 1) % G);
        order[(size_t)d * np + center] = s0 | (s1 << 16);
    }
}

struct LbReq {
    float a, b;
    u32x4_a4 w;
    uint32_t taddr;  // float-encoded LDS offset of the sample's reference texel
};
// the sum of w * dis over the first 2 * kd listed samples of the pixel's window, for one view; `ordp` points at
// the pixel's entry of the first list plane (plane d is np words further: streamed, one word ahead)
template <int BOX, bool FAST>
__device__ __forceinline__ float lb_item(const Problem *__restrict__ P, gptr_bytes magic_base,
                                         const float *__restrict__ H, const float *__restrict__ tp0, int tw,
                                         const float *__restrict__ lut, int px, int py,
                                         const uint32_t *__restrict__ ordp, size_t np, int kd, float *lb_short)
{
    // (*lb_short: the sum two samples short of the end -- what the probe workgroups use to judge the length)
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2;
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float centre = tp0[0];
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const float H1 = H[1], H4 = H[4], H7 = H[7];
    const float pxR = (float)(px - R), pyR = (float)(py - R);
    // byte offset of texel (col, row) of the window from tp0: 16 * ((2 row - R) * tw + 2 col - R), as the
    // low bits of the float 2^23 + 2^15 + offset (|offset| < 2^15: ulp 1, bits = 0x4b008000 + offset)
    constexpr uint32_t kTileMagic = 0x4b008000u;
    static_assert(16 * R * (kTileW + 2 * (R + 1) + 1) < 32768, "tile offsets fit the float encoding");
    const float trow = (float)(32 * tw);
    const float tbias = 8421376.0f - (float)(16 * R * (tw + 1));  // 2^23 + 2^15 - 16 R (tw + 1)
    const char *tile_magic = (const char *)tp0 - kTileMagic;

    auto request = [&](float cif, float rif) -> LbReq {
        // window coordinates as view_cost_pipe forms them (exact small integers)
        const float qx = __builtin_fmaf(cif, 2.0f, pxR), qy = __builtin_fmaf(rif, 2.0f, pyR);
        const float X0 = __builtin_fmaf(H[0], qx, H[2]);
        const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
        const float Z0 = __builtin_fmaf(H[6], qx, H[8]);
        const float X = __builtin_fmaf(H1, qy, X0);
        const float Y = __builtin_fmaf(H4, qy, Y0);
        const float Z = __builtin_fmaf(H7, qy, Z0);
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        LbReq r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
        const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
        r.w = *(gptr_u32x4)(magic_base + off);
        r.taddr = __float_as_uint(__builtin_fmaf(rif, trow, __builtin_fmaf(cif, 32.0f, tbias)));
        return r;
    };
    auto reduce = [&](const LbReq &cur, float acc) -> float {
        const float4 t4 = *reinterpret_cast<const float4 *>(tile_magic + cur.taddr);
        const float colorDis = __builtin_fabsf(t4.x - centre);
        const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
        const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
        const float colDiff = t4.w - tp5.sc;
        const float gradX = t4.y - tp5.gx2;
        const float gradY = t4.z - tp5.gy2;
        const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
        const float colDis = min_abs_nc(colDiff, tau_color);
        const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
        return __builtin_fmaf(w, dis, acc);
    };

    typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
    const gptr_u32 op = (gptr_u32)ordp;
    float lb = 0.0f, prev = 0.0f;
    const uint32_t w0 = op[0];
    uint32_t nxt = op[kd > 1 ? np : 0];
    LbReq r0 = request(ub0(w0), ub1(w0)), r1 = request(ub2(w0), ub3(w0));
    for (int d = 0; d < kd; d++) {
        prev = lb;
        // samples 2d + 2 and 2d + 3 are requested while 2d and 2d + 1 are reduced; the list word after them is
        // on its way (the two requests past the last sample fetch valid, clamped addresses and are dropped)
        const uint32_t cw = nxt;
        nxt = op[(size_t)min(d + 2, kLbDwords - 1) * np];
        LbReq cur = r0;
        r0 = r1;
        r1 = request(ub0(cw), ub1(cw));
        __builtin_amdgcn_sched_barrier(0);
        lb = reduce(cur, lb);
        __builtin_amdgcn_sched_barrier(0);
        cur = r0;
        r0 = r1;
        r1 = request(ub2(cw), ub3(cw));
        __builtin_amdgcn_sched_barrier(0);
        lb = reduce(cur, lb);
        __builtin_amdgcn_sched_barrier(0);
    }
    *lb_short = prev;
    return lb;
}

__device__ __forceinline__ float l1_3(float x, float y, float z);
// lb_item for T = float4 (-color_processing): the per-sample arithmetic of view_cost_c4_loop (three window
// loads and tap sets, l1_norm(float4) reductions, weight table indexed by |dB|+|dG|+|dR|, integer window
// addressing) on the listed samples; `tp0` points at the pixel's own texel in the float4 {B, G, R, 0} tile
template <int BOX, bool FAST>
__device__ __forceinline__ float lb_item_c4(const Problem *__restrict__ P, const ViewCam &vc,
                                            const float *__restrict__ H, const float *__restrict__ tp0, int tw,
                                            const float *__restrict__ lut, int px, int py,
                                            const uint32_t *__restrict__ ordp, size_t np, int kd, float *lb_short)
{
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2;
    const gptr_bytes packed = (gptr_bytes)vc.packed;
    const uint32_t pw = (uint32_t)P->pw;
    const uint32_t xmax = (uint32_t)(P->cols + 2), ymax = (uint32_t)(P->rows + 2);
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
    const gptr_u32 op = (gptr_u32)ordp;
    float lb = 0.0f, prev = 0.0f;
    for (int d = 0; d < kd; d++) {
        prev = lb;
        const uint32_t cw = op[(size_t)d * np];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int ci = (int)((cw >> (16 * e)) & 255u), ri = (int)((cw >> (16 * e + 8)) & 255u);
            const int i = 2 * ci - R, j = 2 * ri - R;
            const float qx = (float)(px + i), qy = (float)(py + j);
            const float X0 = __builtin_fmaf(H[0], qx, H[2]);
            const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
            const float Z0 = __builtin_fmaf(H[6], qx, H[8]);
            const float *tp = tp0 + 4 * (j * tw + i);
            const float4 lv = *reinterpret_cast<const float4 *>(tp);
            const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                            __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
            const float w = lut[(int)S];
            const float X = __builtin_fmaf(H[1], qy, X0);
            const float Y = __builtin_fmaf(H[4], qy, Y0);
            const float Z = __builtin_fmaf(H[7], qy, Z0);
            const float rz = recip<FAST>(Z);
            const float sx = X * rz, sy = Y * rz;
            const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
            const float a = sx - fx0, b = sy - fy0;
            const uint32_t Xw = min(cvt_u32_sat(fx0 + 2.0f), xmax);
            const uint32_t Yw = min(cvt_u32_sat(fy0 + 2.0f), ymax);
            const gptr_bytes base = packed + (Yw * pw + Xw) * 12u;
            const u32x4_a4 q0 = *(gptr_u32x4)(base), q1 = *(gptr_u32x4)(base + 16), q2 = *(gptr_u32x4)(base + 32);
            Taps t[3];  // word 3k+c = column k, channel c
            t[0] = taps_u8(a, b, q0.x, q0.w, q1.z, q2.y);
            t[1] = taps_u8(a, b, q0.y, q1.x, q1.w, q2.z);
            t[2] = taps_u8(a, b, q0.z, q1.y, q2.x, q2.w);
            const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
            const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
            const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
            const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
            const float colDiff = l1_3(lv.x - t[0].sc, lv.y - t[1].sc, lv.z - t[2].sc);
            const float gX = l1_3((right.x - left.x) - t[0].gx2, (right.y - left.y) - t[1].gx2,
                                  (right.z - left.z) - t[2].gx2);
            const float gY = l1_3((down.x - up.x) - t[0].gy2, (down.y - up.y) - t[1].gy2,
                                  (down.z - up.z) - t[2].gy2);
            const float gradDis = min_nc((gX + gY) * 0.0625f, tau_gradient);
            const float colDis = min_nc(colDiff, tau_color);
            const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
            lb = __builtin_fmaf(w, dis, lb);
        }
    }
    *lb_short = prev;
    return lb;
}

template <int BOX, bool U8, bool INTERIOR, bool ET = false>
__device__ __forceinline__ float view_cost(const Problem *__restrict__ P, const ViewCam &vc,
                                           const float *__restrict__ tp0, int tw,
                                           const float *__restrict__ lut, int px, int py, float4 pl,
                                           const Win<BOX> &win, float tau = 0.0f, int *cols_done = nullptr)
{
    float H[9];
    homography(P->rc.K_inv, vc, pl, H);
    const int hr = win.hrad(), vr = win.vrad();
    const bool safe = window_z_safe(H, (float)(px - hr), (float)(px + hr), (float)(py - vr), (float)(py + vr));
    if constexpr (U8) {
        if (P->magic_addr) {
            if constexpr (BOX > 0) {
                if (__all(safe)) return view_cost_pipe<BOX, true, ET>(P, vc, H, tp0, tw, lut, px, py, tau, cols_done);
                return view_cost_pipe<BOX, false, ET>(P, vc, H, tp0, tw, lut, px, py, tau, cols_done);
            } else {
                if (__all(safe))
                    return view_cost_loop<BOX, U8, INTERIOR, true, true>(P, vc, H, tp0, tw, lut, px, py, win);
                return view_cost_loop<BOX, U8, INTERIOR, false, true>(P, vc, H, tp0, tw, lut, px, py, win);
            }
        }
    }
    if (__all(safe)) return view_cost_loop<BOX, U8, INTERIOR, true, false>(P, vc, H, tp0, tw, lut, px, py, win);
    return view_cost_loop<BOX, U8, INTERIOR, false, false>(P, vc, H, tp0, tw, lut, px, py, win);
}

// The same patch cost instantiated for T = float4 (-color_processing, gipuma.cu:1965-1968): every
// image difference is taken per channel and reduced with l1_norm(float4) = mean |.| of x, y, z
// (gipuma.cu:174-179; the float4 operators zero .w, vector_operations.h:9-14).  `tp0` points at the
// lane's own texel in a float4 LDS tile.  U8: weight table indexed by the integer
// |dB|+|dG|+|dR| (766 values), three 16-byte loads for the 4x4x3 window.
__device__ __forceinline__ float l1_3(float x, float y, float z)
{
    return (__builtin_fabsf(x) + __builtin_fabsf(y) + __builtin_fabsf(z)) * 0.3333333f;
}

template <int BOX, bool U8, bool FAST, bool ET = false>
__device__ __forceinline__ float view_cost_c4_loop(const Problem *__restrict__ P, const ViewCam &vc,
                                                   const float *__restrict__ H, const float *__restrict__ tp0,
                                                   int tw, const float *__restrict__ lut, int px, int py,
                                                   const Win<BOX> &win, float tau = 0.0f, int c0 = 0, int c1 = 1 << 20,
                                                   float cost0 = 0.0f, int *cols_run = nullptr)
{
    // (c0, c1, cost0: window columns [c0, c1) only, continuing from the partial sum cost0 -- see
    //  view_cost_pipe_range / refine_two_phase)
    const gptr_f32 img = (gptr_f32)vc.img;
    const uint32_t *__restrict__ packed = vc.packed;
    const uint32_t pw = (uint32_t)P->pw;
    const uint32_t xmax = (uint32_t)(P->cols + 2), ymax = (uint32_t)(P->rows + 2);
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const float colsf = (float)cols, rowsf = (float)rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient, gamma = P->gamma;
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    const int hr = win.hrad(), vr = win.vrad();
    float cost = cost0;
    float qx = (float)(px - hr + 2 * c0);
    int col = c0;
    for (int i = -hr + 2 * c0; i <= hr && col < c1; i += 2, qx += 2.0f) {
        const float X0 = __builtin_fmaf(H[0], qx, H[2]);
        const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
        const float Z0 = __builtin_fmaf(H[6], qx, H[8]);
        float qy = (float)(py - vr);
        for (int j = -vr; j <= vr; j += 2, qy += 2.0f) {
            const float *tp = tp0 + 4 * (j * tw + i);
            const float4 lv = *reinterpret_cast<const float4 *>(tp);
            float w;
            if (U8) {
                const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                                __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
                w = lut[(int)S];
            } else {
                w = exp_model(-l1_3(lv.x - centre.x, lv.y - centre.y, lv.z - centre.z) / gamma);
            }
            const float X = __builtin_fmaf(H[1], qy, X0);
            const float Y = __builtin_fmaf(H[4], qy, Y0);
            const float Z = __builtin_fmaf(H[7], qy, Z0);
            const float rz = recip<FAST>(Z);
            const float sx = X * rz, sy = Y * rz;
            const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
            const float a = sx - fx0, b = sy - fy0;
            Taps t[3];
            if (U8) {
                const uint32_t Xw = min(cvt_u32_sat(fx0 + 2.0f), xmax);
                const uint32_t Yw = min(cvt_u32_sat(fy0 + 2.0f), ymax);
                const uint32_t off = (Yw * pw + Xw) * 12u;
                const gptr_bytes base = (gptr_bytes)packed + off;
                const u32x4_a4 q0 = *(gptr_u32x4)(base), q1 = *(gptr_u32x4)(base + 16), q2 = *(gptr_u32x4)(base + 32);
                // word 3k+c = column k, channel c
                t[0] = taps_u8(a, b, q0.x, q0.w, q1.z, q2.y);
                t[1] = taps_u8(a, b, q0.y, q1.x, q1.w, q2.z);
                t[2] = taps_u8(a, b, q0.z, q1.y, q2.x, q2.w);
            } else {
                const int ix = (int)__builtin_fminf(__builtin_fmaxf(fx0, -2.0f), colsf);
                const int iy = (int)__builtin_fminf(__builtin_fmaxf(fy0, -2.0f), rowsf);
                const int c0 = 4 * clampi(ix - 1, 0, cols - 1), c1 = 4 * clampi(ix, 0, cols - 1);
                const int c2 = 4 * clampi(ix + 1, 0, cols - 1), c3 = 4 * clampi(ix + 2, 0, cols - 1);
                const int r0 = clampi(iy - 1, 0, rows - 1) * pitch, r1 = clampi(iy, 0, rows - 1) * pitch;
                const int r2 = clampi(iy + 1, 0, rows - 1) * pitch, r3 = clampi(iy + 2, 0, rows - 1) * pitch;
#pragma unroll
                for (int c = 0; c < 3; c++)
                    t[c] = taps12(a, b, img[r0 + c1 + c], img[r0 + c2 + c], img[r1 + c0 + c], img[r1 + c1 + c],
                                  img[r1 + c2 + c], img[r1 + c3 + c], img[r2 + c0 + c], img[r2 + c1 + c],
                                  img[r2 + c2 + c], img[r2 + c3 + c], img[r3 + c1 + c], img[r3 + c2 + c]);
            }
            const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
            const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
            const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
            const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
            const float colDiff = l1_3(lv.x - t[0].sc, lv.y - t[1].sc, lv.z - t[2].sc);
            const float gX = l1_3((right.x - left.x) - t[0].gx2, (right.y - left.y) - t[1].gx2,
                                  (right.z - left.z) - t[2].gx2);
            const float gY = l1_3((down.x - up.x) - t[0].gy2, (down.y - up.y) - t[1].gy2,
                                  (down.z - up.z) - t[2].gy2);
            const float gradDis = min_nc((gX + gY) * 0.0625f, tau_gradient);
            const float colDis = min_nc(colDiff, tau_color);
            const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
            cost = __builtin_fmaf(w, dis, cost);
        }
        col++;
        if (ET && __all(cost >= tau)) break;  // early termination, see multiview_cost
    }
    if (cols_run) *cols_run += col - c0;
    return cost;
}

template <int BOX, bool U8, bool ET = false>
__device__ __forceinline__ float view_cost_c4(const Problem *__restrict__ P, const ViewCam &vc,
                                              const float *__restrict__ tp0, int tw,
                                              const float *__restrict__ lut, int px, int py, float4 pl,
                                              const Win<BOX> &win, float tau = 0.0f)
{
    float H[9];
    homography(P->rc.K_inv, vc, pl, H);
    const int hr = win.hrad(), vr = win.vrad();
    const bool safe = window_z_safe(H, (float)(px - hr), (float)(px + hr), (float)(py - vr), (float)(py + vr));
    if (__all(safe)) return view_cost_c4_loop<BOX, U8, true, ET>(P, vc, H, tp0, tw, lut, px, py, win, tau);
    return view_cost_c4_loop<BOX, U8, false, ET>(P, vc, H, tp0, tw, lut, px, py, win, tau);
}

// Accumulation of the per-view costs of pmCostMultiview_cu (gipuma.cu:771-805), shared by the
// pixel-per-lane and the column-per-lane evaluation.  COMBINE_REG: best-N with n_best <= 4 keeps the
// four smallest view costs in registers (a sorting-network insert per view, same values and the
// same ascending summation order as sort_small + the loop at :781-797); otherwise the view costs go
// through a per-lane LDS column `cv` and the literal insertion sort.
template <bool COMBINE_REG>
struct ViewCombiner {
    int numValid = 0;
    float b0 = kMaxCost, b1 = kMaxCost, b2 = kMaxCost, b3 = kMaxCost;
    __device__ __forceinline__ void add(float c, int v, float *cv)
    {
        if (c < kMaxCost)
            numValid++;
        else
            c = kMaxCost;
        if (COMBINE_REG) {
            float t = c, lo;
            lo = __builtin_fminf(b0, t); t = __builtin_fmaxf(b0, t); b0 = lo;
            lo = __builtin_fminf(b1, t); t = __builtin_fmaxf(b1, t); b1 = lo;
            lo = __builtin_fminf(b2, t); t = __builtin_fmaxf(b2, t); b2 = lo;
            b3 = __builtin_fminf(b3, t);
        } else {
            // sort_small (gipuma.cu:684-693) as an online insertion into the lane's column
            int j = v;
            for (; j >= 1 && c < cv[(j - 1) * kThreads]; j--) cv[j * kThreads] = cv[(j - 1) * kThreads];
            cv[j * kThreads] = c;
        }
    }
    // m-th smallest value so far (m = 1..4; COMBINE_REG only): b0 <= b1 <= b2 <= b3, so it is the
    // largest of the first m; spelled with min/max so that the registers are not spilled to an array
    __device__ __forceinline__ float kth(int m) const
    {
        const float inf = __builtin_inff();
        const float s1 = m >= 2 ? inf : -inf, s2 = m >= 3 ? inf : -inf, s3 = m >= 4 ? inf : -inf;
        return __builtin_fmaxf(__builtin_fmaxf(b0, __builtin_fminf(b1, s1)),
                               __builtin_fmaxf(__builtin_fminf(b2, s2), __builtin_fminf(b3, s3)));
    }
    __device__ __forceinline__ float finish(const Problem *__restrict__ P, int n, const float *cv) const
    {
        float cost = 0.0f;
        int numConsidered = 0;
        if (COMBINE_REG) {
            const int numBest = min(numValid, P->n_best);
            if (numBest > 0) cost = cost + b0;
            if (numBest > 1) cost = cost + b1;
            if (numBest > 2) cost = cost + b2;
            if (numBest > 3) cost = cost + b3;
            numConsidered = numBest;
        } else {
            int numBest = numValid;
            if (P->cost_comb == 1) numBest = min(numBest, P->n_best);  // COMB_BEST_N
            if (P->cost_comb == 3) numBest = n;                        // COMB_GOOD
            const float costThresh = (n > 0 ? cv[0] : 0.0f) * P->good_factor;
            for (int i = 0; i < numBest; i++) {
                numConsidered++;
                float c = cv[i * kThreads];
                if (P->cost_comb == 3) c = __builtin_fminf(c, costThresh);
                cost = cost + c;
            }
        }
        cost = cost / ((float)numConsidered);
        if (numConsidered < 1) cost = kMaxCost;
        if (cost != cost || cost > kMaxCost || cost < 0) cost = kMaxCost;
        return cost;
    }
};

// pmCostMultiview_cu, gipuma.cu:720-806
//
// Early termination (ET; best-N with n_best <= 4 on packed gray planes, enabled by the host through
// Problem::et_enable only when the parameters make every view cost finite and < MAXCOST, so that
// numValid == n_sel for every plane).  Work reduction that cannot change a result:
//   A view cost is a sum of terms w*dis >= 0 accumulated by fmaf, so its partial sums never
//   decrease: a view stopped early leaves a LOWER BOUND l_v <= c_v.  Let m = min(n_sel, n_best),
//   b[0..m-1] the m smallest values seen so far (exact costs and lower bounds alike) and
//   tau = min(b[m-1], thr).  A view is abandoned -- by the whole wavefront, after a window column --
//   once every lane's partial sum has reached its own tau.  At the end F' = mean of b[0..m-1] is a
//   lower bound of the exact result F (the m smallest of elementwise smaller values, summed in the
//   same order; rounding is monotone), and
//     * if b[m-1] < thr, no abandoned view is among the m smallest: one abandoned against b[m-1] had
//       m values at or below it already, one abandoned against thr is >= thr > b[m-1].  The m smallest
//       are exact and every other view is >= b[m-1]: F' == F bit for bit;
//     * else if F' >= bound (the cost the candidate must beat): F >= F', the candidate is rejected
//       either way and its cost is never stored;
//     * else the caller re-evaluates with thr = infinity (first case).
//   thr = infinity leaves only the value-exact rule; Problem::et_theta scales thr = theta * bound.
template <int BOX, bool U8, bool INTERIOR, bool COMBINE_REG, int CH, bool ET = false>
__device__ __forceinline__ float multiview_cost(const Problem *__restrict__ P, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, float *cv, int px,
                                                int py, float4 pl, const Win<BOX> &win, bool et_on = false,
                                                float thr = 0.0f, float *kth_out = nullptr, int *cols_done = nullptr)
{
    static_assert(!ET || (COMBINE_REG && U8 && (CH == 4 || BOX > 0)), "ET: register combiner on packed 8-bit planes");
    const int n = P->n_sel;
    const int m = min(n, P->n_best);
    ViewCombiner<COMBINE_REG> comb;
    for (int v = 0; v < n; v++) {
        float c;
        if constexpr (CH == 4 && ET) {
            const float tau = et_on ? __builtin_fminf(comb.kth(m), thr) : __builtin_inff();
            c = view_cost_c4<BOX, U8, true>(P, P->view[v], tp0, tw, lut, px, py, pl, win, tau);
        } else if constexpr (CH == 4) {
            c = view_cost_c4<BOX, U8>(P, P->view[v], tp0, tw, lut, px, py, pl, win);
        } else if constexpr (ET) {
            const float tau = et_on ? __builtin_fminf(comb.kth(m), thr) : __builtin_inff();
            c = view_cost<BOX, U8, INTERIOR, true>(P, P->view[v], tp0, tw, lut, px, py, pl, win, tau, cols_done);
        } else {
            c = view_cost<BOX, U8, INTERIOR>(P, P->view[v], tp0, tw, lut, px, py, pl, win);
        }
        comb.add(c, v, cv);
    }
    if constexpr (ET)
        if (kth_out) *kth_out = comb.kth(m);
    return comb.finish(P, n, cv);
}

// ---------------------------------------------------------------------------------------------
// Column-per-lane evaluation, used for the first iterations (planes still random).
//
// With one lane per pixel, the 64 lanes of a window load sit in 64 different cache lines as long as
// neighbouring pixels hold unrelated planes, and the vector L1 needs two clocks per distinct
// 128-byte line (scripts/ubench/l1_window_rate.hip): launches 0-3 run at 2.2x their VALU bound.
// Here 8 consecutive lanes evaluate ONE (pixel, plane) pair, lane c taking window column c: at
// each of the N row steps the 8 lanes sample the same plane at points 2 pixels apart, i.e. windows
// that share one to three lines, and a wavefront (8 pairs) touches ~20 lines per load instead of
// ~64.  Every sample is computed by the same instruction sequence as in view_cost_pipe.  The
// reference's summation order (columns outer, rows inner, one fmaf per sample into a single
// accumulator, gipuma.cu:633-676) is kept by a relay: each lane stores the N (w, dis) pairs of its
// column; in relay step c every lane re-runs its N fmafs starting from the value its left
// neighbour produced in step c-1, so after step c lane c holds the exact prefix over columns 0..c
// (the other lanes' values are never used).  N*(N+1) extra instructions per N samples per lane --
// irrelevant where the launch is bound by line fills.
// ---------------------------------------------------------------------------------------------
#ifndef PM_COLS_PD
#define PM_COLS_PD 8
#endif
// lanes per (pixel, plane) pair: 8 for windows of up to 8 columns (box <= 15), 16 for up to 16
// (box 25: 13 columns, three lanes of a group shadow the last one); groups never straddle a DPP row
template <int BOX>
__host__ __device__ constexpr int col_group()
{
    return (BOX + 1) / 2 <= 8 ? 8 : 16;
}
template <int BOX>
__host__ __device__ constexpr int col_tasks()  // pairs evaluated concurrently by a workgroup
{
    return kThreads / col_group<BOX>();
}

template <int BOX, bool FAST>
__device__ __forceinline__ float view_cost_cols(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py, int col)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(BOX > 0 && N <= col_group<BOX>(), "one lane per window column");
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float centre = tp0[0];
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)vc.packed - (uintptr_t)kMagicBits);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const float H1 = H[1], H4 = H[4], H7 = H[7];
    const int mycol = col < N ? col : N - 1;  // spare lanes of a smaller box shadow the last column
    const float qx = (float)(px - R + 2 * mycol);
    const float X0 = __builtin_fmaf(H[0], qx, H[2]);
    const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
    const float Z0 = __builtin_fmaf(H[6], qx, H[8]);

    auto request = [&](float qy) -> WinReq {
        const float X = __builtin_fmaf(H1, qy, X0);
        const float Y = __builtin_fmaf(H4, qy, Y0);
        const float Z = __builtin_fmaf(H7, qy, Z0);
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        WinReq r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
        const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
        r.w = *(gptr_u32x4)(magic_base + off);
        return r;
    };

    const float qy0 = (float)(py - R);
    const float *tcol = tp0 + 4 * (-R * tw - R + 2 * mycol);  // texel (column, -R) of the window
    float wgt[N], dis[N];
    // PD window requests in flight (these launches wait on L2 misses, and the kernel has registers
    // to spare below its 3-wavefront budget)
    constexpr int PDmax = N > 8 ? 4 : PM_COLS_PD;  // (13 samples per column: keep the registers for wgt/dis)
    constexpr int PD = PDmax < N ? PDmax : N;
    WinReq req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = request(qy0 + (float)(2 * p));
#pragma unroll
    for (int k = 0; k < N; k++) {
        const WinReq cur = req[k % PD];
        if (k + PD < N) req[k % PD] = request(qy0 + (float)(2 * (k + PD)));
        const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
        const float colorDis = __builtin_fabsf(t4.x - centre);
        wgt[k] = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
        const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
        const float colDiff = t4.w - tp5.sc;
        const float gradX = t4.y - tp5.gx2;
        const float gradY = t4.z - tp5.gy2;
        const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
        const float colDis = min_abs_nc(colDiff, tau_color);
        dis[k] = __builtin_fmaf(alpha, gradDis, oma * colDis);
    }
    // relay: after step c, lane c of the group holds the sum over columns 0..c in reference order
    float out = 0.0f;
#pragma unroll
    for (int c = 0; c < N; c++) {
        // lane i takes lane i-1's value: DPP row_shr:1 (groups of 8 never straddle a row of 16)
        float acc = c == 0 ? 0.0f
                           : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out), 0x111, 0xf, 0xf, false));
#pragma unroll
        for (int k = 0; k < N; k++) acc = __builtin_fmaf(wgt[k], dis[k], acc);
        out = acc;
    }
    return out;  // exact in lane N-1 of the group
}

// view_cost_cols for T = float4 (-color_processing): lane c evaluates window column c by the
// arithmetic of view_cost_c4_loop (three 16-byte window loads and tap sets per sample, l1_norm(float4)
// reductions, weight table indexed by |dB|+|dG|+|dR|, integer window addressing), the relay keeps the
// reference's summation order.  `tp0` points at the pixel's own texel in the float4 {B, G, R, 0} tile.
struct WinReq3 {
    float a, b;
    u32x4_a4 q0, q1, q2;
};
template <int BOX, bool FAST>
__device__ __forceinline__ float view_cost_cols_c4(const Problem *__restrict__ P, const ViewCam &vc,
                                                   const float *__restrict__ H, const float *__restrict__ tp0,
                                                   int tw, const float *__restrict__ lut, int px, int py, int col)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(BOX > 0 && N <= col_group<BOX>(), "one lane per window column");
    const gptr_bytes packed = (gptr_bytes)vc.packed;
    const uint32_t pw = (uint32_t)P->pw;
    const uint32_t xmax = (uint32_t)(P->cols + 2), ymax = (uint32_t)(P->rows + 2);
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    const float H1 = H[1], H4 = H[4], H7 = H[7];
    const int mycol = col < N ? col : N - 1;
    const float qx = (float)(px - R + 2 * mycol);
    const float X0 = __builtin_fmaf(H[0], qx, H[2]);
    const float Y0 = __builtin_fmaf(H[3], qx, H[5]);
    const float Z0 = __builtin_fmaf(H[6], qx, H[8]);

    auto request = [&](float qy) -> WinReq3 {
        const float X = __builtin_fmaf(H1, qy, X0);
        const float Y = __builtin_fmaf(H4, qy, Y0);
        const float Z = __builtin_fmaf(H7, qy, Z0);
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        WinReq3 r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const uint32_t Xw = min(cvt_u32_sat(fx0 + 2.0f), xmax);
        const uint32_t Yw = min(cvt_u32_sat(fy0 + 2.0f), ymax);
        const gptr_bytes base = packed + (Yw * pw + Xw) * 12u;
        r.q0 = *(gptr_u32x4)(base);
        r.q1 = *(gptr_u32x4)(base + 16);
        r.q2 = *(gptr_u32x4)(base + 32);
        return r;
    };

    const float qy0 = (float)(py - R);
    const float *tcol = tp0 + 4 * (-R * tw - R + 2 * mycol);  // texel (column, -R) of the window
    float wgt[N], dis[N];
#ifndef PM_COLS_C4_PD
#define PM_COLS_C4_PD 3
#endif
    constexpr int PD = PM_COLS_C4_PD < N ? PM_COLS_C4_PD : N;  // window requests (three loads each) in flight
    WinReq3 req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = request(qy0 + (float)(2 * p));
#pragma unroll
    for (int k = 0; k < N; k++) {
        const WinReq3 cur = req[k % PD];
        if (k + PD < N) req[k % PD] = request(qy0 + (float)(2 * (k + PD)));
        const float *tp = tcol + 8 * k * tw;
        const float4 lv = *reinterpret_cast<const float4 *>(tp);
        const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                        __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
        wgt[k] = lut[(int)S];
        Taps t[3];  // word 3k+c = column k, channel c
        t[0] = taps_u8(cur.a, cur.b, cur.q0.x, cur.q0.w, cur.q1.z, cur.q2.y);
        t[1] = taps_u8(cur.a, cur.b, cur.q0.y, cur.q1.x, cur.q1.w, cur.q2.z);
        t[2] = taps_u8(cur.a, cur.b, cur.q0.z, cur.q1.y, cur.q2.x, cur.q2.w);
        const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
        const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
        const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
        const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
        const float colDiff = l1_3(lv.x - t[0].sc, lv.y - t[1].sc, lv.z - t[2].sc);
        const float gX = l1_3((right.x - left.x) - t[0].gx2, (right.y - left.y) - t[1].gx2,
                              (right.z - left.z) - t[2].gx2);
        const float gY = l1_3((down.x - up.x) - t[0].gy2, (down.y - up.y) - t[1].gy2,
                              (down.z - up.z) - t[2].gy2);
        const float gradDis = min_nc((gX + gY) * 0.0625f, tau_gradient);
        const float colDis = min_nc(colDiff, tau_color);
        dis[k] = __builtin_fmaf(alpha, gradDis, oma * colDis);
    }
    // relay: after step c, lane c of the group holds the sum over columns 0..c in reference order
    float out = 0.0f;
#pragma unroll
    for (int c = 0; c < N; c++) {
        float acc = c == 0 ? 0.0f
                           : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out), 0x111, 0xf, 0xf, false));
#pragma unroll
        for (int k = 0; k < N; k++) acc = __builtin_fmaf(wgt[k], dis[k], acc);
        out = acc;
    }
    return out;  // exact in lane N-1 of the group
}

// pmCostMultiview_cu for one (pixel, plane) pair evaluated by a group of col_group<BOX>() lanes; the
// result is exact in every lane of the group
template <int BOX, bool COMBINE_REG, int CH = 1>
__device__ __forceinline__ float multiview_cost_cols(const Problem *__restrict__ P, const float *__restrict__ tp0,
                                                     int tw, const float *__restrict__ lut, float *cv, int px,
                                                     int py, float4 pl, int col)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    const int n = P->n_sel;
    ViewCombiner<COMBINE_REG> comb;
    constexpr int G = col_group<BOX>();
    const int grp_lane0 = (int)(threadIdx.x & 63u & ~(unsigned)(G - 1));
    const int src_lane = grp_lane0 + (N - 1);
    // the homography of a (plane, view) pair is the same for the lanes of a group: lane c computes
    // it for view vb + c (the literal arithmetic of homography()), the lanes then pass them round
    for (int vb = 0; vb < n; vb += G) {
        float Hl[9];
        homography(P->rc.K_inv, P->view[min(vb + col, n - 1)], pl, Hl);
        const int vend = min(vb + G, n);
        for (int v = vb; v < vend; v++) {
            float H[9];
#pragma unroll
            for (int k = 0; k < 9; k++) H[k] = __shfl(Hl[k], grp_lane0 + (v - vb));
            const bool safe = window_z_safe(H, (float)(px - R), (float)(px + R), (float)(py - R), (float)(py + R));
            float c;
            if constexpr (CH == 4) {
                if (__all(safe))
                    c = view_cost_cols_c4<BOX, true>(P, P->view[v], H, tp0, tw, lut, px, py, col);
                else
                    c = view_cost_cols_c4<BOX, false>(P, P->view[v], H, tp0, tw, lut, px, py, col);
            } else if (__all(safe))
                c = view_cost_cols<BOX, true>(P, P->view[v], H, tp0, tw, lut, px, py, col);
            else
                c = view_cost_cols<BOX, false>(P, P->view[v], H, tp0, tw, lut, px, py, col);
            comb.add(__shfl(c, src_lane), v, cv);  // the group's exact value
        }
    }
    return comb.finish(P, n, cv);
}

// ---------------------------------------------------------------------------------------------
// workgroup helpers
// ---------------------------------------------------------------------------------------------
// Workgroup id -> tile coordinates.
//
// Workgroups are dealt round-robin to the 8 XCDs (b % 8), each with its own 4 MB L2.  The tiles
// an XCD works on at the same time (~5 workgroups x 32 CUs) should form a compact 2-D block, so
// that their source-view footprints (tile + window halo, in each of the N views) overlap as much
// as possible in that L2:  (1) every XCD gets one contiguous chunk of tile ids; (2) tile ids run
// column-major inside horizontal bands of ceil(gy/8) tile rows, so consecutive ids are vertical
// neighbours and a run of ~160 ids is a ~17 x 10 tile block, not three full-width rows.
struct TileXY {
    int x, y;
};
__device__ __forceinline__ TileXY tile_of(int b, int gx, int gy, unsigned tune)
{
    const int nblk = gx * gy;
    int t = b;
    if (!(tune & Tune::kNoXcdRemap) && nblk >= 8) {
        const int xcd = b & 7, local = b >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    TileXY o;
    if (tune & Tune::kRowMajorTiles) {
        o.x = t % gx;
        o.y = t / gx;
    } else {
        const int bo = (int)((tune >> 8) & 0x3ffu);          // experiment override of the band height (bits 8..17)
        const int bh = bo ? min(bo, gy) : (gy + 7) >> 3;  // band height in tile rows
        const int band = t / (bh * gx);
        const int h = min(bh, gy - band * bh);   // the last band may be shorter
        const int rem = t - band * bh * gx;
        o.x = rem / h;
        o.y = band * bh + rem % h;
    }
    return o;
}

// stage the reference tile (+halo) and the weight table; the tile holds clamp-to-edge point
// samples exactly like the reference's (gipuma.cu:1393-1402, 1513-1522)
// (PAD: extra texels per tile row, so that lanes two tile rows apart do not share LDS banks)
template <int BOX, int CH, int PAD = 0>
__device__ __forceinline__ void stage_tile(const Problem *__restrict__ P, float *lds, int x0, int y0,
                                           int tile_h, const Win<BOX> &win, bool want_lut)
{
    const int hw = win.halo_w(), hh = win.halo_h();
    const int tw = kTileW + 2 * hw, th = tile_h + 2 * hh;
    const int tws = tw + PAD;  // row stride of the float4 tile
    const gptr_f32 ref = (gptr_f32)P->ref;
    float *tile = lds + lut_size<CH>();  // float4 per texel
    // gray: the scalar image goes to a scratch plane behind the float4 tile first, so that the
    // central differences can be formed once per tile instead of once per sample
    float *plane = tile + 4 * tws * th;
    for (int k = threadIdx.x; k < tw * th; k += kThreads) {
        const int ty = k / tw, tx = k - ty * tw;
        const int gx = clampi(x0 - hw + tx, 0, P->cols - 1);
        const int gy = clampi(y0 - hh + ty, 0, P->rows - 1);
        if (CH == 4) {
            const gptr_f32 s = ref + (gy * P->pitch + 4 * gx);
            *reinterpret_cast<float4 *>(tile + 4 * (ty * tws + tx)) = make_float4(s[0], s[1], s[2], 0.0f);
        } else {
            plane[k] = ref[gy * P->pitch + gx];
        }
    }
    if (want_lut)
        for (int k = threadIdx.x; k < lut_size<CH>(); k += kThreads)
            lds[k] = exp_model(-(CH == 4 ? (float)k * 0.3333333f : (float)k) / P->gamma);
    __syncthreads();
    if (CH == 1) {
        // {I, gx1, gy1, -} with gx1 = I(x+1) - I(x-1), gy1 = I(y+1) - I(y-1): the reference-side
        // terms of pmCostComputation_shared (gipuma.cu:254-259), same fp32 subtractions
        for (int k = threadIdx.x; k < tw * th; k += kThreads) {
            const int ty = k / tw, tx = k - ty * tw;
            float gx1 = 0.0f, gy1 = 0.0f;
            if (tx > 0 && tx < tw - 1 && ty > 0 && ty < th - 1) {
                gx1 = plane[k + 1] - plane[k - 1];
                gy1 = plane[k + tw] - plane[k - tw];
            }
            // .w repeats I so that the per-sample read uses all four dwords: one ds_read_b128 (4 LDS
            // cycles) instead of the ds_read_b96 (8) the compiler picks for three
            *reinterpret_cast<float4 *>(tile + 4 * (ty * tws + tx)) = make_float4(plane[k], gx1, gy1, plane[k]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// the random plane of gipuma_init_cu2 for pixel (px, py): disparity uniform in the range, normal by
// Marsaglia's method flipped towards the camera (gipuma.cu:1019-1034, 148-164, 131-137)
__device__ __forceinline__ float4 random_plane(const Problem *__restrict__ P, int px, int py)
{
    const RefCam &rc = P->rc;
    const uint32_t pre = rng_prefix(P->seed, 0u, (uint32_t)px, (uint32_t)py);
    uint32_t draw = 0;
    const Vec3 view = view_vector(rc, px, py);
    const float disp = between(rng_uniform(pre, draw++), P->min_disp, P->max_disp);
    // rndUnitVectorSphereMarsaglia_cu, gipuma.cu:148-164
    float rx = 1.0f, ry = 1.0f, sum = 2.0f;
    while (sum >= 1.0f) {
        rx = between(rng_uniform(pre, draw++), -1.0f, 1.0f);
        ry = between(rng_uniform(pre, draw++), -1.0f, 1.0f);
        sum = rx * rx + ry * ry;
    }
    const float sq = __builtin_sqrtf(1.0f - sum);
    Vec3 n;
    n.x = 2.0f * rx * sq;
    n.y = 2.0f * ry * sq;
    n.z = 1.0f - 2.0f * sum;
    n = on_hemisphere(n, view);
    const float depth = disp_depth(rc.f, rc.baseline, disp);
    return make_float4(n.x, n.y, n.z, plane_d(rc, n, px, py, depth));
}

// gipuma_init_cu2 (gipuma.cu:996-1051) when GENERATE, else the cost of a given plane field
// (gipuma_initial_cost, :1052-1079).  32x8 tile, one lane per pixel.
template <int BOX, bool U8, bool COMBINE_REG, bool GENERATE, int CH>
__global__ __launch_bounds__(kThreads) void init_kernel(const Problem *__restrict__ P,
                                                        float4 *__restrict__ norm4, float *__restrict__ cost,
                                                        unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Win<BOX> win(P);
    const int gx = (P->cols + kTileW - 1) / kTileW;
    const int gy = (P->rows + kDenseTileH - 1) / kDenseTileH;
    const TileXY txy = tile_of(blockIdx.x, gx, gy, tune);
    const int x0 = txy.x * kTileW, y0 = txy.y * kDenseTileH;
    stage_tile<BOX, CH>(P, lds, x0, y0, kDenseTileH, win, U8);
    const int hw = win.halo_w(), hh = win.halo_h();
    const int tw = kTileW + 2 * hw;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    if (px >= P->cols || py >= P->rows) return;
    const float *tp0 = lds + lut_size<CH>() + ((ly + hh) * tw + (lx + hw)) * 4;
    float *cv = lds + lut_size<CH>() + 4 * tw * (kDenseTileH + 2 * hh) +
                work_floats<CH>(tw * (kDenseTileH + 2 * hh), false) + threadIdx.x;
    const int center = py * P->cols + px;
    float4 pl;
    if (GENERATE) {
        pl = random_plane(P, px, py);
        norm4[center] = pl;
    } else {
        pl = norm4[center];
    }
    float c;
    if (tune & Tune::kNoInterior)
        c = multiview_cost<BOX, U8, false, COMBINE_REG, CH>(P, tp0, tw, lds, cv, px, py, pl, win);
    else
        c = multiview_cost<BOX, U8, true, COMBINE_REG, CH>(P, tp0, tw, lds, cv, px, py, pl, win);
    cost[center] = c;
}

// init_kernel with the column-per-lane evaluation (view_cost_cols): every lane draws / reads the
// plane of its own pixel as above, the 256 planes of the tile go through LDS and are evaluated by
// groups of col_group<BOX>() lanes -- random planes are the worst case for one lane per pixel.
template <int BOX, bool GENERATE, int CH = 1>
__global__ __launch_bounds__(kThreads) void init_cols_kernel(const Problem *__restrict__ P,
                                                             float4 *__restrict__ norm4, float *__restrict__ cost,
                                                             unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Win<BOX> win(P);
    const int gx = (P->cols + kTileW - 1) / kTileW;
    const int gy = (P->rows + kDenseTileH - 1) / kDenseTileH;
    const TileXY txy = tile_of(blockIdx.x, gx, gy, tune);
    const int x0 = txy.x * kTileW, y0 = txy.y * kDenseTileH;
    stage_tile<BOX, CH>(P, lds, x0, y0, kDenseTileH, win, true);
    const int hw = win.halo_w(), hh = win.halo_h();
    const int tw = kTileW + 2 * hw, th = kDenseTileH + 2 * hh;
    const float *tile = lds + lut_size<CH>();
    float *work = lds + lut_size<CH>() + 4 * tw * th;  // the staging plane is dead now
    float4 *candbuf = reinterpret_cast<float4 *>(work);
    float *bres = work + 4 * kThreads;
    float *cv = work + work_floats<CH>(tw * th, false) + threadIdx.x;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = x0 + lx, py = y0 + ly;
    const bool active = px < P->cols && py < P->rows;
    const int center = py * P->cols + px;
    float4 pl = make_float4(0.f, 0.f, -1.f, 1.f);
    if (active) {
        if (GENERATE) {
            pl = random_plane(P, px, py);
            norm4[center] = pl;
        } else {
            pl = norm4[center];
        }
    }
    candbuf[threadIdx.x] = pl;
    __syncthreads();
    constexpr int kColGroup = col_group<BOX>(), kColTasks = col_tasks<BOX>();
    const int grp = threadIdx.x / kColGroup, col = threadIdx.x % kColGroup;
    for (int r = 0; r < kThreads / kColTasks; r++) {
        const int owner = r * kColTasks + grp;
        // pixels outside the image evaluate their dummy plane at the clamped position (never stored)
        const int epx = min(x0 + (owner & 31), P->cols - 1), epy = min(y0 + (owner >> 5), P->rows - 1);
        const float4 ecand = candbuf[owner];
        const float *etp0 = tile + (((epy - y0) + hh) * tw + ((epx - x0) + hw)) * 4;
        const float c = multiview_cost_cols<BOX, false, CH>(P, etp0, tw, lds, cv, epx, epy, ecand, col);
        if (col == 0) bres[owner] = c;
    }
    __syncthreads();
    if (active) cost[center] = bres[threadIdx.x];
}

// One colour of one iteration: the bodies of gipuma_checkerboard_spatialPropClose_cu
// (gipuma.cu:1471-1588), ..._spatialPropFar_cu (:1353-1468) and ..._planeRefinement_cu
// (:1590-1711) in one launch.  Fusing them is result-identical: every pixel of a colour reads only
// its own state and pixels of the OTHER colour (distances 1 and 5 are odd, :1730-1734), which no
// lane of this launch writes.  `stages` selects a subset so the three reference launches can also
// be reproduced one by one.
//
// Work reduction that cannot change a result ("exact skipping"):
//   The cost of a plane at a pixel is a pure function of (pixel, plane).  Of the up-to-8
//   propagation candidates of a pixel, one that is BITWISE equal to
//     (A) the pixel's current plane -- its cost is exactly the stored cost (the state invariant
//         cost[p] == cost(p, plane[p]) holds after init_planes and after every accept; it is NOT
//         assumed after gipuma_hip_set_state, see `trust`), so `c < cost_now` is false; or
//     (D) an earlier candidate of the same pixel -- same cost and same depth test; if the earlier
//         one was accepted then c == cost_now, if it was rejected then c >= cost_then >= cost_now
//     (H) the unchanged plane of a neighbour: if this pixel's colour and the neighbour's colour
//         have been swept strictly alternately with all stages (the host checks the sequence and
//         sets Tune::kHistorySkip), and the neighbour's plane did not change in its last half-sweep
//         (Problem::changed), then this pixel met exactly that plane one half-sweep ago and did
//         not end up with it at a lower cost -- it was rejected against a cost that has only
//         decreased since, or accepted and improved upon
//     (S) a plane this pixel's propagation evaluated before (a ring of its last kSeenRing evaluated
//         candidates, Problem::seen_ring): its cost F is a pure function of (pixel, plane); it was then
//         rejected against a cost that has only decreased since (or for its depth, which is a pure
//         function too), or accepted -- and the pixel's cost has been <= F ever since.  This does not
//         need the state invariant, only that the pixel's cost never increases between the two
//         half-sweeps: the host clears the rings whenever planes are (re-)installed.  A plane that
//         spreads over a patch reaches a pixel that turned it down again and again, through every
//         neighbour that adopts it: 8 % (fifth half-sweep) to 20 % (last) of the remaining candidates
//   can never be accepted (strict <, gipuma.cu:868) and is not evaluated.  On config C the
//   evaluated candidates drop from 8 to 2.7 per pixel by the last half-sweep.  Because the
//   per-wavefront MAXIMUM stays near 8, the surviving (pixel, candidate) pairs of the whole
//   workgroup are compacted -- a wavefront-level scan plus a 4-counter exchange through LDS -- into
//   one task list (owner-major: a pixel's surviving candidates are adjacent) and evaluated 256 at a
//   time by whichever lane is free, which also balances the four wavefronts; the owner lane then replays its accept decisions in the
//   reference order up, down, left, right (distance 1, then 5) from the stored costs.
__device__ __forceinline__ bool same_bits(float4 a, float4 b)
{
    return ((__float_as_uint(a.x) ^ __float_as_uint(b.x)) | (__float_as_uint(a.y) ^ __float_as_uint(b.y)) |
            (__float_as_uint(a.z) ^ __float_as_uint(b.z)) | (__float_as_uint(a.w) ^ __float_as_uint(b.w))) == 0u;
}

// neighbour of candidate slot k (0..3 distance 1, 4..7 distance 5; up, down, left, right) and
// whether the reference's guard lets it be tested (gipuma.cu:1571-1582, 1450-1462)
__device__ __forceinline__ bool neighbour(int k, int px, int py, int rows, int cols, int center, int &nb)
{
    const int dist = k < 4 ? 1 : 5;
    switch (k & 3) {
    case 0: nb = center - dist * cols; return py > dist - 1;
    case 1: nb = center + dist * cols; return py < rows - dist;
    case 2: nb = center - dist; return px > dist - 1;
    default: nb = center + dist; return px < cols - dist;
    }
}

// ---- pieces of a half-sweep shared by sweep_kernel and sweep_cols_kernel ----
// Per-lane context: the lane's pixel inside the workgroup tile, the LDS carve, and the pixel's state.
struct SweepLane {
    int x0, y0, hw, hh, tw;           // tile origin, halo, tile row length (texels)
    int lx, ly, px, py, center;      // pixel inside the tile / in the image
    bool active;                      // inside the image
    const float *tile;                // reference tile (float4 per texel)
    float *bres;                      // [8][256] candidate costs
    unsigned short *btask;            // [2048] tasks: owner tid | slot << 8
    int *wcnt;                        // per-wavefront counters
    float *cv;                        // this lane's column of view costs (generic combiner)
    float4 pl;                        // current plane
    float cst, depth;                 // its cost and depth
    unsigned needmask;                // candidate slots that must be evaluated
    unsigned chg;                     // the plane changed in this half-sweep
    int n_tasks;                      // surviving (pixel, candidate) pairs of the workgroup
    int n_seen;                       // (statistics) candidates of this pixel removed by rule (S)
};

// tile staging, lane -> pixel mapping, state read (gipuma.cu:1527-1530) and the exact skipping rules:
// leaves L.needmask = the candidate slots of this lane's pixel that must be evaluated
template <int BOX, int CH, int PAD = 0>
__device__ __forceinline__ void sweep_read_state(SweepLane &L, const Problem *__restrict__ P, float *lds,
                                                 const float4 *__restrict__ norm4, const float *__restrict__ cost,
                                                 int colour, unsigned stages, unsigned tune, bool want_lut)
{
    const Win<BOX> win(P);
    const RefCam &rc = P->rc;
    const int rows = P->rows, cols = P->cols;
    const int gx = (cols + kTileW - 1) / kTileW;
    const int gy = (rows + kSweepTileH - 1) / kSweepTileH;
    const TileXY txy = tile_of(blockIdx.x, gx, gy, tune);
    L.x0 = txy.x * kTileW;
    L.y0 = txy.y * kSweepTileH;
    stage_tile<BOX, CH, PAD>(P, lds, L.x0, L.y0, kSweepTileH, win, want_lut);
    L.hw = win.halo_w();
    L.hh = win.halo_h();
    L.tw = kTileW + 2 * L.hw + PAD;  // row stride of the tile
    const int th = kSweepTileH + 2 * L.hh;
    L.tile = lds + lut_size<CH>();
    float *work = lds + lut_size<CH>() + 4 * L.tw * th;
    L.cv = work + work_floats<CH>(L.tw * th, true) + threadIdx.x;
    // (gray: the staging plane inside `work` is dead after stage_tile's last barrier)
    L.bres = work;                                                                  // [8][256] costs
    L.btask = reinterpret_cast<unsigned short *>(work + 8 * kThreads);              // [2048] tid | slot << 8
    L.wcnt = reinterpret_cast<int *>(work + 8 * kThreads + (8 * kThreads) / 2);     // [4 waves][8 slots]

    // lane -> pixel: 16 pixels of the colour per tile row; a wavefront covers 4 rows x 32 columns
    L.ly = threadIdx.x >> 4;
    L.lx = 2 * (threadIdx.x & 15) + ((L.ly + colour) & 1);  // tile origin is even in x and y
    L.px = L.x0 + L.lx;
    L.py = L.y0 + L.ly;
    L.active = L.px < cols && L.py < rows;
    L.center = L.py * cols + L.px;
    const bool trust = !(tune & Tune::kUntrustedCosts);

    // read state (gipuma.cu:1527-1530)
    L.pl = make_float4(0.f, 0.f, 0.f, 0.f);
    L.cst = 0.f;
    L.depth = 0.f;
    L.needmask = 0;
    L.chg = 0;
    L.n_tasks = 0;
    L.n_seen = 0;
    const bool history = (tune & Tune::kHistorySkip) != 0;
    if (L.active) {
        const float4 pl = norm4[L.center];
        L.pl = pl;
        L.cst = cost[L.center];
        L.depth = depth_from_plane(rc, pl, L.px, L.py);
        float4 cands[8];
        unsigned valid = 0, needmask = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int nb;
            const bool ok = neighbour(k, L.px, L.py, rows, cols, L.center, nb) && (stages & (k < 4 ? 1u : 2u));
            if (ok) {
                cands[k] = norm4[nb];
                valid |= 1u << k;
            }
        }
        if (tune & Tune::kPushConsume) {
            // the neighbours evaluated their planes for this pixel (pm_push.h) unless rule (H) let them
            // keep silent; a cost that rules (A) / (D) would have skipped is replayed and rejected
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if ((valid >> k) & 1u) {
                    int nb;
                    neighbour(k, L.px, L.py, rows, cols, L.center, nb);
                    if (!history || P->changed[nb] != 0) needmask |= 1u << k;
                }
            }
        } else if (tune & Tune::kNoSkip) {
            needmask = valid;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                bool fresh = (valid >> k) & 1u;
                if (fresh && trust && same_bits(cands[k], pl)) fresh = false;  // (A)
                if (fresh && history) {                                        // (H)
                    int nb;
                    neighbour(k, L.px, L.py, rows, cols, L.center, nb);
                    if (P->changed[nb] == 0) fresh = false;
                }
#pragma unroll
                for (int j = 0; j < k; j++)
                    if (fresh && ((valid >> j) & 1u) && same_bits(cands[k], cands[j])) fresh = false;  // (D)
                if (fresh) needmask |= 1u << k;
            }
            if (P->seen_ring != nullptr && !(tune & Tune::kNoSeen)) {  // (S)
                const size_t np = (size_t)rows * (size_t)cols;
                const unsigned st = P->seen_pos[L.center];
                const unsigned before_seen = needmask;
                const int cnt = (st & 8u) ? kSeenRing : (int)(st & 7u);
#pragma unroll
                for (int a = 0; a < kSeenRing; a++) {
                    if (a < cnt && needmask != 0u) {
                        const float4 e = P->seen_ring[(size_t)a * np + (size_t)L.center];
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (((needmask >> k) & 1u) && same_bits(cands[k], e)) needmask &= ~(1u << k);
                    }
                }
                L.n_seen = __popc(before_seen) - __popc(needmask);
                unsigned pos = st & 7u, full = st & 8u;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if ((needmask >> k) & 1u) {
                        P->seen_ring[(size_t)pos * np + (size_t)L.center] = cands[k];
                        pos = (pos + 1u) & 7u;
                        if (pos == 0u) full = 8u;
                    }
                }
                P->seen_pos[L.center] = (unsigned char)(pos | full);
            }
        }
        L.needmask = needmask;
    }
}

// sweep_read_state + the workgroup task list
template <int BOX, int CH, int PAD = 0>
__device__ __forceinline__ void sweep_setup(SweepLane &L, const Problem *__restrict__ P, float *lds,
                                            const float4 *__restrict__ norm4, const float *__restrict__ cost,
                                            int colour, unsigned stages, unsigned tune, bool want_lut)
{
    sweep_read_state<BOX, CH, PAD>(L, P, lds, norm4, cost, colour, stages, tune, want_lut);
    if (tune & Tune::kPushConsume) return;  // nothing to evaluate: L.n_tasks == 0, the replay reads Problem::push_cost
    // Workgroup task list.  Two orders, same set of tasks (the order cannot change a result: a task
    // is a pure function of (pixel, plane) and its cost lands in bres[slot][owner]):
    //  * source-major (default): tasks that evaluate the SAME plane -- the plane of other-colour
    //    pixel q is a candidate of q+-1 and q+-5 in x and y -- are adjacent, so the lanes of a
    //    wavefront that share a plane read source windows a few pixels apart, i.e. the same cache
    //    lines, even while the planes themselves are still random.  A divergent window load costs
    //    the vector L1 two clocks per distinct 128-byte line (scripts/ubench/l1_window_rate.hip);
    //    the first half-sweeps are bound by exactly that.  Built as a counting sort keyed by q's
    //    position in the tile extended by the 5-pixel propagation reach.
    //  * owner-major (Tune::kOwnerMajorTasks): a pixel's surviving candidates adjacent, pixels in
    //    lane order: a wavefront-level inclusive scan of the per-lane counts.
    int n_tasks;
    const unsigned needmask = L.needmask;
    const int lx = L.lx, ly = L.ly, wave = threadIdx.x >> 6;
    float *bres = L.bres;
    unsigned short *btask = L.btask;
    int *wcnt = L.wcnt;
    if (tune & Tune::kOwnerMajorTasks) {
        const int cnt = __popc(needmask);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if ((int)(threadIdx.x & 63) >= d) incl += up;
        }
        if ((threadIdx.x & 63) == 63) wcnt[wave] = incl;
        __syncthreads();
        const int c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
        int pos0 = incl - cnt + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        n_tasks = c0 + c1 + c2 + c3;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((needmask >> k) & 1u) btask[pos0++] = (unsigned short)(threadIdx.x | (k << 8));
        __syncthreads();
    } else {
        constexpr int kReach = 5, kExtW = kTileW + 2 * kReach, kExtH = kSweepTileH + 2 * kReach;
        constexpr int kCells = kExtW * kExtH, kPerLane = (kCells + kThreads - 1) / kThreads;
        static_assert(kCells <= 8 * kThreads, "the histogram aliases bres");
        int *hist = reinterpret_cast<int *>(bres);  // bres is not written before the first round
        for (int c = threadIdx.x; c < kCells; c += kThreads) hist[c] = 0;
        __syncthreads();
        unsigned ranks = 0;  // 3 bits per slot: at most 8 tasks share a source pixel
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if ((needmask >> k) & 1u) {
                const int dist = k < 4 ? 1 : 5;
                const int qx = lx + ((k & 3) == 2 ? -dist : (k & 3) == 3 ? dist : 0) + kReach;
                const int qy = ly + ((k & 3) == 0 ? -dist : (k & 3) == 1 ? dist : 0) + kReach;
                ranks |= (unsigned)atomicAdd(&hist[qy * kExtW + qx], 1) << (3 * k);
            }
        }
        __syncthreads();
        // exclusive prefix sum of the histogram, kPerLane consecutive cells per lane
        int loc[kPerLane];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < kPerLane; e++) {
            const int c = threadIdx.x * kPerLane + e;
            loc[e] = c < kCells ? hist[c] : 0;
            sum += loc[e];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if ((int)(threadIdx.x & 63) >= d) incl += up;
        }
        if ((threadIdx.x & 63) == 63) wcnt[wave] = incl;
        __syncthreads();
        const int c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
        int run = incl - sum + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        n_tasks = c0 + c1 + c2 + c3;
#pragma unroll
        for (int e = 0; e < kPerLane; e++) {
            const int c = threadIdx.x * kPerLane + e;
            if (c < kCells) hist[c] = run;
            run += loc[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if ((needmask >> k) & 1u) {
                const int dist = k < 4 ? 1 : 5;
                const int qx = lx + ((k & 3) == 2 ? -dist : (k & 3) == 3 ? dist : 0) + kReach;
                const int qy = ly + ((k & 3) == 0 ? -dist : (k & 3) == 1 ? dist : 0) + kReach;
                btask[hist[qy * kExtW + qx] + (int)((ranks >> (3 * k)) & 7u)] =
                    (unsigned short)(threadIdx.x | (k << 8));
            }
        }
        __syncthreads();  // also orders the last reads of `hist` before bres is written
    }
    L.n_tasks = n_tasks;
}

// replay: spatialPropagation_cu's accept test (gipuma.cu:865-872) in slot order, by the owner lane,
// from the costs the rounds left in bres
__device__ __forceinline__ void sweep_replay(SweepLane &L, const Problem *__restrict__ P,
                                             const float4 *__restrict__ norm4, bool pushed = false)
{
    const RefCam &rc = P->rc;
    const size_t np = (size_t)P->rows * (size_t)P->cols;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((L.needmask >> k) & 1u) {
            int nb;
            neighbour(k, L.px, L.py, P->rows, P->cols, L.center, nb);
            const float4 cand = norm4[nb];
            const float c = pushed ? P->push_cost[(size_t)k * np + (size_t)L.center] : L.bres[k * kThreads + threadIdx.x];
            const float d_new = depth_from_plane(rc, cand, L.px, L.py);
            if (d_new >= rc.depth_min && d_new <= rc.depth_max && c < L.cst) {  // :829-830, :868
                L.depth = d_new;
                L.pl = cand;
                L.cst = c;
                L.chg = 1;
            }
        }
    }
}

// planeRefinement_cu + getRndDispAndUnitVector_cu, gipuma.cu:928-994, 890-927
struct RefineDraws {
    int nref;  // number of refinement steps: deltaZ = max_disp/2, /10 ... >= 0.01 (:958-959)
    Vec3 view;
    uint32_t pre, draw;
    float deltaN, deltaZ;
};
__device__ __forceinline__ void refine_init(RefineDraws &R, const Problem *__restrict__ P, unsigned stages)
{
    R.nref = 0;
    if (stages & 4u)
        for (float dz = P->max_disp / 2.0f; dz >= 0.01f; dz = dz / 10.0f) R.nref++;
    R.view.x = R.view.y = R.view.z = 0.f;
    R.pre = 0;
    R.draw = 0;
    R.deltaN = 1.0f;
    R.deltaZ = P->max_disp / 2.0f;
}
// after the propagation accepts: the refine kernel re-derives the depth (:1660) and seeds its draws
__device__ __forceinline__ void refine_begin(RefineDraws &R, SweepLane &L, const Problem *__restrict__ P,
                                             uint32_t phase)
{
    if (R.nref > 0 && L.active) {
        L.depth = depth_from_plane(P->rc, L.pl, L.px, L.py);
        R.view = view_vector(P->rc, L.px, L.py);
        R.pre = rng_prefix(P->seed, phase, (uint32_t)L.px, (uint32_t)L.py);
    }
}
// the candidate of the current step for an active lane (four draws)
__device__ __forceinline__ float4 refine_candidate(RefineDraws &R, const SweepLane &L,
                                                   const Problem *__restrict__ P, float &d_new)
{
    const RefCam &rc = P->rc;
    const float min_disp = P->min_disp, max_disp = P->max_disp;
    const float disp = disp_depth(rc.f, rc.baseline, L.depth);
    const float minDelta = -__builtin_fminf(R.deltaZ, min_disp + disp);  // sic, :909
    const float maxDelta = __builtin_fminf(R.deltaZ, max_disp - disp);
    const float u0 = rng_uniform(R.pre, R.draw++);
    const float u1 = rng_uniform(R.pre, R.draw++);
    const float u2 = rng_uniform(R.pre, R.draw++);
    const float u3 = rng_uniform(R.pre, R.draw++);
    const float dz = between(u0, minDelta, maxDelta);
    const float dispOut = __builtin_fminf(__builtin_fmaxf(disp + dz, min_disp), max_disp);
    d_new = disp_depth(rc.f, rc.baseline, dispOut);
    Vec3 n;
    n.x = L.pl.x + between(u1, -R.deltaN, R.deltaN);
    n.y = L.pl.y + between(u2, -R.deltaN, R.deltaN);
    n.z = L.pl.z + between(u3, -R.deltaN, R.deltaN);
    n = on_hemisphere(normalize3(n), R.view);
    return make_float4(n.x, n.y, n.z, plane_d(rc, n, L.px, L.py, d_new));
}
__device__ __forceinline__ void refine_next_step(RefineDraws &R)
{
    R.deltaN = R.deltaN / 4.0f;
    R.deltaZ = R.deltaZ / 10.0f;
}
// pixel of task-list owner `owner` (a lane id of this workgroup)
__device__ __forceinline__ void owner_pixel(const SweepLane &L, int owner, int colour, int &olx, int &oly)
{
    oly = owner >> 4;
    olx = 2 * (owner & 15) + ((oly + colour) & 1);
}

// Two-phase evaluation of one refinement step of a workgroup (performance only; packed 8-bit planes
// -- gray with float-encoded offsets or colour --, compile-time box, register combiner).
//
// multiview_cost's bounded evaluation leaves a view when the SLOWEST of 64 lanes has reached its
// bound; the average lane gets there after a third of the window (scripts/exp/et_stats.py), the
// slowest of 64 after more than half.  Here the unit of work is the (candidate, view) ITEM:
//   phase 1  every lane evaluates the first g0 window columns of every view of its own candidate
//            (the partial sums go to LDS); an item whose partial sum has not reached tau = thr
//            survives, and the survivors of the workgroup are appended to one list;
//   phase 2  the list is dealt out 64 items per wavefront: a lane picks up an item of any pixel --
//            its plane and bound from LDS, the homography recomputed by the literal arithmetic of
//            homography() -- and continues the sum from column g0, the wavefront leaving when all
//            of its items have reached their bounds;
//   combine  every lane collects the values of its candidate's views -- exact costs, or lower
//            bounds >= thr -- in view order through the same ViewCombiner.
// Views are handled in groups of kTpViews (LDS space).  The values are those of view_cost_pipe
// stopped at a column boundary, and the bound is thr alone (not min(b[m-1], thr)), so the three
// cases of multiview_cost's proof apply unchanged: kth < thr -> exact; else F' >= bound -> rejected;
// else the caller calls again with thr = inf, g0 = 0 for the lanes left open (every view of theirs
// becomes an item and is summed in full).  Which lane evaluates an item cannot matter: an item is
// a pure function of (pixel, plane, view, columns).
// All lanes of the workgroup must call this (barriers); `seq` counts the groups processed so far
// in this launch (the two item counters are used alternately: the one not in use is cleared while
// nobody touches it).
// (one (candidate, view) item over the window columns [c0, c1); `vc` may differ per lane)
template <int BOX, int CH, bool FAST>
__device__ __forceinline__ float tp_item(const Problem *__restrict__ P, const ViewCam &vc, const float *__restrict__ H,
                                         const float *__restrict__ tp0, int tw, const float *__restrict__ lut, int px,
                                         int py, int c0, int c1, float a, float tau, int *cols_run)
{
    if constexpr (CH == 4) {
        const Win<BOX> win(P);
        return view_cost_c4_loop<BOX, true, FAST, true>(P, vc, H, tp0, tw, lut, px, py, win, tau, c0, c1, a, cols_run);
    } else {
        const gptr_bytes base = (gptr_bytes)((uintptr_t)vc.packed - (uintptr_t)kMagicBits);
        return view_cost_pipe_range<BOX, FAST>(P, base, H, tp0, tw, lut, px, py, c0, c1, a, tau, cols_run);
    }
}

// `lbk` > 0 (gray): phase 1 is the lower-bound prefilter instead -- the sum over the pixel's lbk heaviest
// window samples (lb_item, list `ord`); an item it decides passes its bound on, the others run their
// exact chain from column 0 in phase 2.
template <int BOX, int CH>
__device__ __forceinline__ float refine_two_phase(const Problem *__restrict__ P, const SweepLane &L, float *work,
                                                  const float *__restrict__ lut, int colour, bool valid, float4 cand,
                                                  float thr, int g0, float *kth_out, int &seq, int *cols_run,
                                                  int *items_left, int lbk, const uint32_t *ordp,
                                                  int *items_short = nullptr)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    float *accv = work + TpLayout::acc;
    float4 *tplane = reinterpret_cast<float4 *>(work + TpLayout::plane);
    float *ttau = work + TpLayout::tau;
    unsigned short *items = reinterpret_cast<unsigned short *>(work + TpLayout::items);
    int *cnt = reinterpret_cast<int *>(work + TpLayout::cnt);
    const int n = P->n_sel, m = min(n, P->n_best);
    const int tid = threadIdx.x;
    const float ninf = -__builtin_inff();
    const float my_tau = valid ? thr : ninf;  // lanes without a candidate never hold a wavefront back
    tplane[tid] = cand;
    ttau[tid] = thr;
    ViewCombiner<true> comb;
    for (int vb = 0; vb < n; vb += kTpViews, seq++) {
        const int ve = min(vb + kTpViews, n);
        int *ctr = cnt + (seq & 1);
        // ---- phase 1 ----
        for (int v = vb; v < ve; v++) {
            float a = 0.0f;
            bool alive = valid;
            if (lbk > 0) {
                float H[9];
                homography(P->rc.K_inv, P->view[v], cand, H);
                const bool safe = window_z_safe(H, (float)(L.px - R), (float)(L.px + R), (float)(L.py - R), (float)(L.py + R));
                const float *tp0 = L.tile + ((L.ly + L.hh) * L.tw + (L.lx + L.hw)) * 4;
                const size_t np = (size_t)P->rows * (size_t)P->cols;
                float lb, lbs;
                if constexpr (CH == 4) {
                    if (__all(safe))
                        lb = lb_item_c4<BOX, true>(P, P->view[v], H, tp0, L.tw, lut, L.px, L.py, ordp, np, lbk >> 1, &lbs);
                    else
                        lb = lb_item_c4<BOX, false>(P, P->view[v], H, tp0, L.tw, lut, L.px, L.py, ordp, np, lbk >> 1, &lbs);
                } else {
                    const gptr_bytes base = (gptr_bytes)((uintptr_t)P->view[v].packed - (uintptr_t)kMagicBits);
                    if (__all(safe))
                        lb = lb_item<BOX, true>(P, base, H, tp0, L.tw, lut, L.px, L.py, ordp, np, lbk >> 1, &lbs);
                    else
                        lb = lb_item<BOX, false>(P, base, H, tp0, L.tw, lut, L.px, L.py, ordp, np, lbk >> 1, &lbs);
                }
                if (cols_run) *cols_run += 1 + (lbk + N - 1) / N;
                const float bound = lb * kLbShrink;  // <= the reference's chain value (see lb_item)
                const bool dead = bound >= thr && lb >= kLbFloor;
                if (items_short)  // (probe workgroups: items that two samples fewer would have left open)
                    *items_short += (int)__popcll(__ballot(valid && !(lbs * kLbShrink >= thr && lbs >= kLbFloor)));
                a = dead ? bound : 0.0f;
                alive = valid && !dead;
            } else if (g0 > 0) {
                float H[9];
                homography(P->rc.K_inv, P->view[v], cand, H);
                const bool safe = window_z_safe(H, (float)(L.px - R), (float)(L.px + R), (float)(L.py - R), (float)(L.py + R));
                const float *tp0 = L.tile + ((L.ly + L.hh) * L.tw + (L.lx + L.hw)) * 4;
                if (__all(safe))
                    a = tp_item<BOX, CH, true>(P, P->view[v], H, tp0, L.tw, lut, L.px, L.py, 0, g0, 0.0f, my_tau, cols_run);
                else
                    a = tp_item<BOX, CH, false>(P, P->view[v], H, tp0, L.tw, lut, L.px, L.py, 0, g0, 0.0f, my_tau, cols_run);
                if (cols_run) *cols_run += 1;  // (homography and set-up, as in view_cost_pipe's count)
                alive = valid && !(a >= thr);
            }
            accv[(v - vb) * kThreads + tid] = a;
            const unsigned long long bal = __ballot(alive);
            if (bal != 0ull) {
                int first = 0;
                if ((tid & 63) == 0) first = atomicAdd(ctr, (int)__popcll(bal));
                first = __builtin_amdgcn_readfirstlane(first);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (alive) items[first + rank] = (unsigned short)(((v - vb) << 8) | tid);
            }
        }
        __syncthreads();
        const int n_items = *ctr;
        if (tid == 0) cnt[(seq + 1) & 1] = 0;
        if (items_left) *items_left += n_items;  // (the same number in every lane)
        // ---- phase 2 ----
        for (int first = 0; first < n_items; first += kThreads) {
            const int wave_first = first + (tid & ~63);
            if (wave_first >= n_items) continue;  // (wave-uniform)
            const int i = first + tid;
            const bool have = i < n_items;
            const unsigned it = items[have ? i : wave_first];  // spare lanes shadow the wavefront's first item
            const int t = (int)(it & 255u), vl = (int)(it >> 8), v = vb + vl;
            const float4 pl = tplane[t];
            const float tau_i = have ? ttau[t] : ninf;
            float a = accv[vl * kThreads + t];
            int olx, oly;
            owner_pixel(L, t, colour, olx, oly);
            const int epx = L.x0 + olx, epy = L.y0 + oly;
            const float *tp0 = L.tile + ((oly + L.hh) * L.tw + (olx + L.hw)) * 4;
            float H[9];
            homography(P->rc.K_inv, P->view[v], pl, H);
            const bool safe = window_z_safe(H, (float)(epx - R), (float)(epx + R), (float)(epy - R), (float)(epy + R));
            const int c_from = lbk > 0 ? 0 : g0;
            if (__all(safe))
                a = tp_item<BOX, CH, true>(P, P->view[v], H, tp0, L.tw, lut, epx, epy, c_from, N, a, tau_i, cols_run);
            else
                a = tp_item<BOX, CH, false>(P, P->view[v], H, tp0, L.tw, lut, epx, epy, c_from, N, a, tau_i, cols_run);
            if (cols_run) *cols_run += 1;
            if (have) accv[vl * kThreads + t] = a;
        }
        __syncthreads();
        // ---- combine (view order, as multiview_cost) ----
        for (int v = vb; v < ve; v++) comb.add(accv[(v - vb) * kThreads + tid], v, nullptr);
    }
    *kth_out = comb.kth(m);
    return comb.finish(P, n, nullptr);
}

// One of the first three refinement steps of a workgroup by (candidate, view) items
// (refine_two_phase), where the previous half-sweep's probe workgroups (every 16th) found that
// bounding the step pays; returns false -- nothing done -- where it does not.  All lanes of the
// workgroup must call it (the decision is uniform over the workgroup); `cand` / `d_new`: the lane's
// candidate (refine_candidate) where do_eval.
template <int BOX, int CH>
__device__ __forceinline__ bool refine_step_items(const Problem *__restrict__ P, SweepLane &L, const float *lds,
                                                  int colour, uint32_t phase, int step, bool do_eval, float4 cand,
                                                  float d_new, int &tp_seq)
{
    const bool probe = (blockIdx.x & 15u) == 0u;
    const unsigned *seen = P->et_stat + ((phase + 2u) % 3u) * kEtSlot + 4 * step;
    // (no measurement yet -- the previous half-sweep ran the column-per-lane kernel --: assume
    //  it pays from the fifth half-sweep on)
    const bool pays = P->et_enable > 1 ||  // (tests: every workgroup bounds every step)
                      (seen[0] > 0u ? (unsigned long long)seen[1] * 100ull <= (unsigned long long)seen[0] * 85ull
                                    : phase >= 5u);
    if (!(probe || pays)) return false;  // (uniform over the workgroup)
    constexpr int Nc = (BOX + 1) / 2;
    // phase-1 length: 3/8 of the window, then one column more / less than the previous
    // half-sweep's probes used if more than 40 % / fewer than 10 % of their items survived it
    int g0 = (3 * Nc + 4) / 8;
    if (P->tp_g0 > 0) {
        g0 = min(P->tp_g0, Nc);
    } else if (seen[3] > 0u) {
        g0 = (int)P->et_stat[((phase + 2u) % 3u) * kEtSlot + 12 + step];
        if ((unsigned long long)seen[2] * 100ull > (unsigned long long)seen[3] * 40ull) g0++;
        if ((unsigned long long)seen[2] * 100ull < (unsigned long long)seen[3] * 10ull) g0--;
        g0 = max(2, min(g0, (Nc + 1) / 2 + 1));
    }
    const int g0_used = g0;
    // lower-bound prefilter (gray): length from Problem::lb_k, or two samples more / fewer than the previous
    // half-sweep's probes used if more than 12 % / fewer than 3 % of their items survived it
    int lbk = 0;
    const uint32_t *ordp = P->worder;
    if (ordp != nullptr) {
        constexpr int kLbFirst = Nc * Nc / 4 < kLbMax ? (Nc * Nc / 4) & ~1 : kLbMax;  // no measurement yet: a quarter of the window
        if (P->lb_k > 0) {
            lbk = min(P->lb_k & ~1, kLbMax);
        } else if (P->lb_k == 0) {
            lbk = kLbFirst;
            if (seen[3] > 0u) {
                // an open item costs about kLbOpen samples (homography again + its chain up to the bound);
                // the last two samples of the previous probes' prefilter paid if they closed more than
                // 2 / kLbOpen of the items; two more are tried while more than 8 % stay open
                constexpr unsigned long long kLbOpen = 45;
                const unsigned prev_k = P->et_stat[((phase + 2u) % 3u) * kEtSlot + 16 + step];
                const unsigned long long open_k = seen[2], open_short = P->et_stat[((phase + 2u) % 3u) * kEtSlot + 20 + step];
                lbk = (int)prev_k;
                if (prev_k > 0u) {
                    if (open_short >= open_k && (open_short - open_k) * kLbOpen < 2ull * seen[3])
                        lbk -= 2;
                    else if (open_k * 100ull > (unsigned long long)seen[3] * 8ull)
                        lbk += 2;
                }
                lbk = max(4, min(lbk, kLbMax));
            }
        }
        ordp += L.active ? (size_t)L.center : 0;
    }
    const int lbk_used = lbk;
    int items_left = 0, items_short = 0, n_redo = 0;
    float thr = P->et_theta[step] * L.cst;
    bool need = do_eval;
    int cols_run = 0;
    float c = 0.0f;
    for (int pass = 0; pass < 2; pass++) {
        float kth;
        const float cc = refine_two_phase<BOX, CH>(P, L, L.bres, lds, colour, need, cand, thr, g0, &kth, tp_seq,
                                                   probe ? &cols_run : nullptr, pass == 0 ? &items_left : nullptr,
                                                   lbk, ordp, probe && pass == 0 ? &items_short : nullptr);
        const bool open = need && kth >= thr && cc < L.cst;
        if (need && !open) c = cc;
        need = open;
        thr = __builtin_inff();
        g0 = 0;
        lbk = 0;
        if (P->dbg != nullptr) n_redo += __syncthreads_count(need);
        if (!__syncthreads_or(need)) break;
    }
    if (P->dbg != nullptr && threadIdx.x == 0) {
        unsigned long long *d = P->dbg + (size_t)(phase & 63u) * kDbgSlots;
        atomicAdd(&d[kDbgItemsOpen], (unsigned long long)items_left);
        atomicAdd(&d[kDbgRedo], (unsigned long long)n_redo);
    }
    if (P->dbg != nullptr) {
        const unsigned n_cand = (unsigned)__popcll(__ballot(do_eval));
        if ((threadIdx.x & 63u) == 0u) {
            unsigned long long *d = P->dbg + (size_t)(phase & 63u) * kDbgSlots;
            atomicAdd(&d[kDbgCands], (unsigned long long)n_cand);
            atomicAdd(&d[kDbgItems], (unsigned long long)n_cand * (unsigned)P->n_sel);
        }
    }
    if (probe) {
        const unsigned n_cand = (unsigned)__popcll(__ballot(do_eval));
        if ((threadIdx.x & 63u) == 0u) {
            unsigned *mine = P->et_stat + (phase % 3u) * kEtSlot + 4 * step;
            atomicAdd(&mine[0], (unsigned)(P->n_sel * (Nc + 1)));
            atomicAdd(&mine[1], (unsigned)cols_run);
            atomicAdd(&mine[3], n_cand * (unsigned)P->n_sel);
            if (threadIdx.x == 0) {
                atomicAdd(&mine[2], (unsigned)items_left);
                P->et_stat[(phase % 3u) * kEtSlot + 12 + step] = (unsigned)g0_used;
                P->et_stat[(phase % 3u) * kEtSlot + 16 + step] = (unsigned)lbk_used;
            }
            if ((threadIdx.x & 63u) == 0u && items_short > 0) {  // (per wavefront: ballots of its own lanes)
                atomicAdd(&P->et_stat[(phase % 3u) * kEtSlot + 20 + step], (unsigned)items_short);
            }
        }
    }
    if (do_eval && c < L.cst) {  // refinement has no depth-range test, :986
        L.depth = d_new;
        L.pl = cand;
        L.cst = c;
        L.chg = 1;
    }
    return true;
}

// (the packed-gray instantiations are held at 128 VGPRs = 4 wavefronts per SIMD)
#ifndef PM_SWEEP_WG
#define PM_SWEEP_WG 3  // workgroups per CU the packed-gray sweep kernel is compiled for (3: 168 VGPRs, no spills; 4: 128 VGPRs, measured level)
#endif
template <int BOX, bool U8, bool COMBINE_REG, bool INTERIOR, int CH>
__global__ __launch_bounds__(kThreads, U8 ? (CH == 4 ? 4 : PM_SWEEP_WG) : 1) void sweep_kernel(const Problem *__restrict__ P,
                                                         float4 *__restrict__ norm4, float *__restrict__ cost,
                                                         int colour, uint32_t phase, unsigned stages,
                                                         unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Win<BOX> win(P);
    const int rows = P->rows, cols = P->cols;
    SweepLane L;
    sweep_setup<BOX, CH>(L, P, lds, norm4, cost, colour, stages, tune, U8);
    const int prop_rounds = (L.n_tasks + kThreads - 1) / kThreads;
    if (P->dbg != nullptr) {
        unsigned long long *d = P->dbg + (size_t)(phase & 63u) * kDbgSlots;
        if (threadIdx.x == 0) atomicAdd(&d[kDbgTasks], (unsigned long long)L.n_tasks);
        int ns = L.n_seen;  // (one atomic per wavefront: per-lane atomics on one address would distort the timing)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ns += __shfl_xor(ns, o);
        if ((threadIdx.x & 63u) == 0u && ns) atomicAdd(&d[kDbgSeen], (unsigned long long)ns);
    }
    RefineDraws R;
    refine_init(R, P, stages);
    constexpr bool ET = U8 && COMBINE_REG && INTERIOR && (CH == 4 || BOX > 0);
    const bool et_on = ET && P->et_enable && !(tune & Tune::kNoEarlyExit);
    if (et_on && blockIdx.x == 0 && threadIdx.x < kEtSlot) P->et_stat[((phase + 1u) % 3u) * kEtSlot + threadIdx.x] = 0u;
    int tp_seq = 0;  // refine_two_phase: groups of views processed so far
    if constexpr (ET && BOX > 0)
        if (threadIdx.x < 2) reinterpret_cast<int *>(L.bres + TpLayout::cnt)[threadIdx.x] = 0;  // (barriers follow)

    // One loop, one call site of the cost function: rounds [0, prop_rounds) evaluate compacted
    // propagation tasks (possibly of another lane's pixel), then the owner replays its accepts,
    // then rounds [prop_rounds, prop_rounds + nref) are the lane's own refinement steps.
    for (int r = 0; r <= prop_rounds + R.nref; r++) {
        if (r == prop_rounds) {
            __syncthreads();  // every wavefront runs the same number of rounds, so this is uniform
            sweep_replay(L, P, norm4, (tune & Tune::kPushConsume) != 0);
            refine_begin(R, L, P, phase);
            if constexpr (ET && BOX > 0)
                if (et_on) __syncthreads();  // refine_two_phase reuses the candidate costs the replay has just read
        }
        if (r == prop_rounds + R.nref) break;

        bool do_eval;
        float4 cand = make_float4(0.f, 0.f, -1.f, 1.f);
        int epx = L.px, epy = L.py, slot = 0, owner = threadIdx.x;
        float d_new = 0.f;
        if (r < prop_rounds) {
            const int pos = r * kThreads + threadIdx.x;
            do_eval = pos < L.n_tasks;
            if (do_eval) {
                const unsigned t = L.btask[pos];
                owner = (int)(t & 255u);
                slot = (int)(t >> 8);
                int olx, oly;
                owner_pixel(L, owner, colour, olx, oly);
                epx = L.x0 + olx;
                epy = L.y0 + oly;
                int nb;
                neighbour(slot, epx, epy, rows, cols, epy * cols + epx, nb);
                cand = norm4[nb];
            }
        } else {
            do_eval = L.active;
            if (do_eval) cand = refine_candidate(R, L, P, d_new);
            refine_next_step(R);
        }
        if constexpr (ET && BOX > 0) {
            if (et_on && r >= prop_rounds && r - prop_rounds < 3 && !(tune & Tune::kNoTwoPhase) &&
                refine_step_items<BOX, CH>(P, L, lds, colour, phase, r - prop_rounds, do_eval, cand, d_new, tp_seq))
                continue;
        }
        if (do_eval) {
            const float *etp0 = L.tile + (((epy - L.y0) + L.hh) * L.tw + ((epx - L.x0) + L.hw)) * 4;
            float c = 0.0f;
            if constexpr (ET) {
                // refinement candidates only have to beat L.cst: bound the evaluation by theta * L.cst
                // and redo the (rare) lanes whose outcome the bound leaves open (see multiview_cost).
                // Propagation tasks keep only the value-exact rule (their costs are stored).
                // Where candidates are often accepted (wide disparity ranges, early sweeps) the redo
                // costs more than the bound saves: a wavefront that had to redo leaves a note and
                // evaluates that step unbounded in its next few half-sweeps.
                float thr = __builtin_inff();
                unsigned char *hint = nullptr;
                int hv = 0, step = -1;
                const bool probe = (blockIdx.x & 15u) == 0u;
                // (only the first three refinement steps: later ones -- a wide disparity range has up to
                //  six -- are perturbations so small that the candidate usually ties with the plane)
                if (et_on && r >= prop_rounds && r - prop_rounds < 3) {
                    step = r - prop_rounds;
                    const unsigned *seen = P->et_stat + ((phase + 2u) % 3u) * kEtSlot + 4 * step;
                    const bool pays = P->et_enable > 1 ||  // (tests: every workgroup bounds every step)
                                  (seen[0] > 0u && (unsigned long long)seen[1] * 100ull <= (unsigned long long)seen[0] * 85ull);
                    if (probe) {
                        thr = P->et_theta[step] * L.cst;
                    } else if (pays) {
                        hint = P->et_hint + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 3 + step;
                        hv = __builtin_amdgcn_readfirstlane((int)*hint);  // low nibble: half-sweeps to sit out; high: level
                        if ((hv & 15) == 0) thr = P->et_theta[step] * L.cst;
                    }
                }
                bool need = true, redone = false;
                int cols_done = 0;
                for (int pass = 0; pass < 2; pass++) {
                    if (need) {
                        float kth;
                        const float cc = multiview_cost<BOX, U8, INTERIOR, COMBINE_REG, CH, true>(
                            P, etp0, L.tw, lds, L.cv, epx, epy, cand, win, et_on, thr, &kth,
                            probe && step >= 0 ? &cols_done : nullptr);
                        const bool open = kth >= thr && cc < L.cst;
                        if (open) {
                            thr = __builtin_inff();
                        } else {
                            c = cc;
                            need = false;
                        }
                    }
                    if (!__any(need)) break;
                    redone = true;
                }
                if (probe && step >= 0 && (threadIdx.x & 63u) == 0u) {
                    unsigned *mine = P->et_stat + (phase % 3u) * kEtSlot + 4 * step;
                    atomicAdd(&mine[0], (unsigned)(P->n_sel * ((BOX + 1) / 2 + 1)));
                    atomicAdd(&mine[1], (unsigned)cols_done);
                }
                if (hint && (redone || hv > 0)) {
                    // exponential back-off: a redo raises the level and sits out 2^level - 1 half-sweeps,
                    // a bounded step that went through lowers it
                    const int level = hv >> 4;
                    int nv;
                    if (redone) {
                        const int nl = min(level + 1, 4);
                        nv = (nl << 4) | ((1 << nl) - 1);
                    } else if (hv & 15) {
                        nv = hv - 1;
                    } else {
                        nv = max(level - 1, 0) << 4;
                    }
                    *hint = (unsigned char)nv;
                }
            } else {
                c = multiview_cost<BOX, U8, INTERIOR, COMBINE_REG, CH>(P, etp0, L.tw, lds, L.cv, epx, epy, cand, win);
            }
            if (r < prop_rounds) {
                L.bres[slot * kThreads + owner] = c;
            } else if (c < L.cst) {  // refinement has no depth-range test, :986
                L.depth = d_new;
                L.pl = cand;
                L.cst = c;
                L.chg = 1;
            }
        }
    }

    // write back (gipuma.cu:1585-1587): 16 B + 4 B per active pixel (+ the history flag)
    if (L.active) {
        cost[L.center] = L.cst;
        norm4[L.center] = L.pl;
        P->changed[L.center] = (unsigned char)(L.chg | ((tune & Tune::kAccumChanged) ? P->changed[L.center] : 0u));
    }
}

// The same half-sweep with the column-per-lane evaluation (see view_cost_cols): state, candidate
// selection, task list, accept replay and refinement candidates are computed per pixel by its owner
// lane exactly as in sweep_kernel (the shared helpers above); only the cost evaluations are done by
// groups of col_group<BOX>() lanes, col_tasks<BOX>() (pixel, plane) pairs at a time, exchanging planes and costs
// through LDS.  Gray packed planes with float-encoded offsets and a compile-time box only (the host
// uses it for box 15, whose 8 columns fill a group of 8, and for box 25: 13 of 16 lanes).
template <int BOX, bool COMBINE_REG, int CH = 1>
__global__ __launch_bounds__(kThreads) void sweep_cols_kernel(const Problem *__restrict__ P,
                                                              float4 *__restrict__ norm4, float *__restrict__ cost,
                                                              int colour, uint32_t phase, unsigned stages,
                                                              unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = P->rows, cols = P->cols;
    SweepLane L;
    sweep_setup<BOX, CH>(L, P, lds, norm4, cost, colour, stages, tune, true);
    RefineDraws R;
    refine_init(R, P, stages);

    // propagation: kColTasks tasks per round, one group of lanes each
    constexpr int kColGroup = col_group<BOX>(), kColTasks = col_tasks<BOX>();
    const int grp = threadIdx.x / kColGroup, col = threadIdx.x % kColGroup;
    const int prop_rounds_c = (L.n_tasks + kColTasks - 1) / kColTasks;
    for (int r = 0; r < prop_rounds_c; r++) {
        const int pos = r * kColTasks + grp;
        const bool have = pos < L.n_tasks;
        const unsigned t = L.btask[have ? pos : 0];
        const int owner = (int)(t & 255u), slot = (int)(t >> 8);
        int olx, oly;
        owner_pixel(L, owner, colour, olx, oly);
        const int epx = L.x0 + olx, epy = L.y0 + oly;
        int nb;
        neighbour(slot, epx, epy, rows, cols, epy * cols + epx, nb);
        const float4 cand = norm4[nb];
        const float *etp0 = L.tile + ((oly + L.hh) * L.tw + (olx + L.hw)) * 4;
        const float c = multiview_cost_cols<BOX, COMBINE_REG, CH>(P, etp0, L.tw, lds, L.cv, epx, epy, cand, col);
        if (have && col == 0) L.bres[slot * kThreads + owner] = c;
    }
    __syncthreads();
    sweep_replay(L, P, norm4, (tune & Tune::kPushConsume) != 0);
    refine_begin(R, L, P, phase);
    // refinement steps: the owner draws its candidate, groups evaluate all 256, the owner accepts
    float4 *candbuf = reinterpret_cast<float4 *>(L.btask);  // the task list is dead now (same 4 KB)
    for (int step = 0; step < R.nref; step++) {
        float4 cand = make_float4(0.f, 0.f, -1.f, 1.f);
        float d_new = 0.f;
        if (L.active) cand = refine_candidate(R, L, P, d_new);
        refine_next_step(R);
        __syncthreads();  // the previous step's reads of bres / candbuf are done
        candbuf[threadIdx.x] = cand;
        __syncthreads();
        for (int r = 0; r < kThreads / kColTasks; r++) {
            const int owner = r * kColTasks + grp;
            int olx, oly;
            owner_pixel(L, owner, colour, olx, oly);
            // pixels outside the image (ragged last tile) evaluate their dummy plane at the clamped
            // position: harmless, never read back
            const int epx = min(L.x0 + olx, cols - 1), epy = min(L.y0 + oly, rows - 1);
            const float4 ecand = candbuf[owner];
            const float *etp0 = L.tile + (((epy - L.y0) + L.hh) * L.tw + ((epx - L.x0) + L.hw)) * 4;
            const float c = multiview_cost_cols<BOX, COMBINE_REG, CH>(P, etp0, L.tw, lds, L.cv, epx, epy, ecand, col);
            if (col == 0) L.bres[owner] = c;
        }
        __syncthreads();
        if (L.active) {
            const float c = L.bres[threadIdx.x];
            if (c < L.cst) {  // refinement has no depth-range test, :986
                L.depth = d_new;
                L.pl = cand;
                L.cst = c;
                L.chg = 1;
            }
        }
    }

    // write back (gipuma.cu:1585-1587)
    if (L.active) {
        cost[L.center] = L.cst;
        norm4[L.center] = L.pl;
        P->changed[L.center] = (unsigned char)(L.chg | ((tune & Tune::kAccumChanged) ? P->changed[L.center] : 0u));
    }
}

// gipuma_compute_disp, gipuma.cu:1080-1103
__global__ __launch_bounds__(kThreads) void finalize_kernel(const Problem *__restrict__ P,
                                                            float4 *__restrict__ norm4,
                                                            const float *__restrict__ cost)
{
    const int n = P->rows * P->cols;
    const int center = blockIdx.x * kThreads + threadIdx.x;
    if (center >= n) return;
    const int py = center / P->cols, px = center - py * P->cols;
    const float4 pl = norm4[center];
    Vec3 v = {pl.x, pl.y, pl.z};
    const Vec3 w = matvec(P->rc.R_orig_inv, v);
    float depth = 0.0f;
    if (cost[center] != kMaxCost) depth = depth_from_plane(P->rc, pl, px, py);
    norm4[center] = make_float4(w.x, w.y, w.z, depth);
}

}  // namespace pm
