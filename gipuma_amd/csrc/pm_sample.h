// pm_sample.h -- the per-sample arithmetic of the patch cost, in ONE place for every kernel family: the warp of a window
// sample through the homography (getCorrespondingPoint_cu, gipuma.cu:207-217; matvecmul4noz / vecdiv4, config.h:150-162,
// 44-47), the five bilinear taps (gipuma.cu:251-253), the truncated differences and `dis` (:254-272) and the weighted
// accumulation (:274, :672).  Part of the device code of the PatchMatch path (pm_device.h).
//
// PM_MODEL selects the numerical model of a translation unit, as a mask of the choices the reference's source leaves to
// nvcc and the texture unit (the same mask as the CPU restatement's flavour switch):
//     bit 0 (1)  every tap its own bilinear fetch at the coordinates the source writes, (pt.x +- 1 + 0.5f, pt.y + 0.5f),
//                with its own fraction, and the taps in the reference's order (gipuma.cu:251-253)  -- else model M1: one
//                4x4 window, the centre tap's fractions, the +-1 differences taken on the texels (taps12);
//     bit 1 (2)  x / z and y / z as correctly rounded IEEE quotients (config.h:44-47)  -- else M2: x * (1/z);
//     bit 2 (4)  H (x, y, 1), dis and cost + w dis as UNFUSED multiply-adds in source order (config.h:150-162,
//                gipuma.cu:272-274, 672)  -- else M3: the fmaf nesting of round 1-5.
//   7 = the reference's own operation order: GIPUMA_HIP_FLAG_LITERAL (gipuma_hip_literal.hip), bit-identical to the
//       reference's code compiled for the CPU with fp32 filter weights;
//   6 = the default since round 6: what separates it from 7 is the order of the taps only, which measurably does not matter
//       (DESIGN.md 4); config B 99.97 % of the pixels inside the north_star tolerance of the reference's code where model 0
//       has 94.6 %;
//   0 = the model of rounds 1-5 plus its PM_APPROX shortcuts: GIPUMA_HIP_FLAG_FAST (gipuma_hip_fast.hip).
// Whatever the model, every kernel family computes a sample by the SAME expressions, so the work-sharing kernels (push,
// plane-keyed, column-per-lane, prefilter, bounded evaluation) stay exact: dis(q, H, view) is a pure function in each.
#pragma once
#include "pm_core.h"

#if PM_LITERAL
#undef PM_MODEL
#define PM_MODEL 7
#elif PM_APPROX
#undef PM_MODEL
#define PM_MODEL 0
#elif !defined(PM_MODEL)
#define PM_MODEL 6
#endif

#if (PM_MODEL & 1) && !PM_LITERAL
#error "the literal taps (PM_MODEL bit 0) are built through PM_LITERAL = 1 (gipuma_hip_literal.hip): tex2d_literal lives under it"
#endif

namespace pm {

constexpr int kModel = PM_MODEL;
constexpr bool kLitTaps = (kModel & 1) != 0, kExactDiv = (kModel & 2) != 0, kUnfused = (kModel & 4) != 0;
static_assert(!kExactDiv || kUnfused, "the proof of the fast exact quotient (window_div_safe) is written for the unfused warp");
static_assert(!(PM_APPROX && kModel != 0), "the approximate shortcuts belong to model 0");

// ---------------------------------------------------------------------------------------------
// warp: H (qx, qy, 1)
// ---------------------------------------------------------------------------------------------
struct WarpCol {  // what a window column (fixed qx) contributes
    float X0, Y0, Z0;
};
struct WarpRow {  // the coefficients the rest needs
    float H1, H4, H7, H2, H5, H8;
};
struct Warped {
    float X, Y, Z;
};
__device__ __forceinline__ WarpCol warp_col(const float *__restrict__ H, float qx)
{
    WarpCol c;
    if (kUnfused) {  // m0 * x (+ m1 * y + m2 below), config.h:150-162
        c.X0 = H[0] * qx;
        c.Y0 = H[3] * qx;
        c.Z0 = H[6] * qx;
    } else {
        c.X0 = __builtin_fmaf(H[0], qx, H[2]);
        c.Y0 = __builtin_fmaf(H[3], qx, H[5]);
        c.Z0 = __builtin_fmaf(H[6], qx, H[8]);
    }
    return c;
}
__device__ __forceinline__ WarpRow warp_row(const float *__restrict__ H)
{
    WarpRow r;
    r.H1 = H[1]; r.H4 = H[4]; r.H7 = H[7];
    r.H2 = H[2]; r.H5 = H[5]; r.H8 = H[8];
    return r;
}
__device__ __forceinline__ Warped warp_point(const WarpCol &c, const WarpRow &r, float qy)
{
    Warped p;
    if (kUnfused) {  // (m0 x + m1 y) + m2, left to right
        p.X = (c.X0 + r.H1 * qy) + r.H2;
        p.Y = (c.Y0 + r.H4 * qy) + r.H5;
        p.Z = (c.Z0 + r.H7 * qy) + r.H8;
    } else {
        p.X = __builtin_fmaf(r.H1, qy, c.X0);
        p.Y = __builtin_fmaf(r.H4, qy, c.Y0);
        p.Z = __builtin_fmaf(r.H7, qy, c.Z0);
    }
    return p;
}

// x / z, y / z (vecdiv4, config.h:44-47).  FAST: the caller has proven (window_div_safe) that the cheap sequence gives the
// bits of the IEEE result on every sample of its window.
//   model bit 1 clear (M2): x * (1/z) with the correctly rounded 1/z -- rcp_newton, 3 instructions;
//   model bit 1 set: the correctly rounded QUOTIENT from that reciprocal by one Markstein step,
//       r = RN(1/z),  q = RN(x r),  q' = RN(q + RN(x - q z) r)       (x - q z is exact in an fma)
//     = RN(x / z) for every pair of fp32 significands (all 2^46 of them compared with the IEEE division on an MI355X:
//     gipuma_hip_selftest_quotient, profiles/r06_selftest_quotient.txt), hence -- the sequence is homogeneous in the two
//     exponents -- for every x, z for which nothing under- or overflows on the way: 9 instructions for both quotients where
//     two IEEE divisions take 20.
template <bool FAST>
__device__ __forceinline__ void warp_divide(const Warped &p, float &sx, float &sy)
{
    if (kExactDiv) {
        if (FAST) {
            const float r = rcp_correct(p.Z);
            const float qx = p.X * r, qy = p.Y * r;
            sx = __builtin_fmaf(__builtin_fmaf(-qx, p.Z, p.X), r, qx);
            sy = __builtin_fmaf(__builtin_fmaf(-qy, p.Z, p.Y), r, qy);
        } else {
            sx = p.X / p.Z;
            sy = p.Y / p.Z;
        }
    } else {
        const float rz = recip<FAST>(p.Z);
        sx = p.X * rz;
        sy = p.Y * rz;
    }
}

// Does the cheap division give the IEEE bits on EVERY sample q in [qx0, qx1] x [qy0, qy1]?
// X, Y, Z of warp_point are monotone in qx and in qy (each operation rounds monotonically), so over the box they lie
// between their four corner values.
//   M2 (reciprocal only): rcp_newton is the correctly rounded 1/z for biased exponents 1..252 (exhaustive:
//     gipuma_hip_selftest_reciprocal); required: all Z of one sign, 2^-100 <= |Z| <= 2^100.
//   exact quotient: additionally nothing may under- or overflow in q = x r, in x - q z, or in the result:
//     2^-40 <= |Z| <= 2^40 and |X|, |Y| <= 2^60 (corners), and X, Y either 0 or >= 2^-60 in magnitude.  The last holds
//     for the unfused warp whenever |H2|, |H5| >= 2^-36:  X = RN(S + H2) with S = RN(H0 qx + H1 qy);  if S and -H2 are
//     within a factor 2 the sum is exact (Sterbenz) and a multiple of ulp(H2) / 2, so 0 or >= 2^-24 |H2| / 2; otherwise
//     |X| >= |H2| / 2 (1 - 2^-24).  Then q >= 2^-100 is normal, x - q z is a multiple of ulp(q) ulp(z) >= 2^-47 |x| / 4
//     >= 2^-109: representable, and the theorem's conditions hold.  (X = 0: q = q' = 0 of either sign; the sign of a
//     zero coordinate reaches no result: floor, the fraction and the window address are the same.)
__device__ __forceinline__ bool window_div_safe(const float *__restrict__ H, float qx0, float qx1, float qy0, float qy1)
{
#if PM_APPROX
    return true;  // (the approx flavour's reciprocal is v_rcp_f32 everywhere: nothing to prove)
#endif
    const WarpRow r = warp_row(H);
    const WarpCol c0 = warp_col(H, qx0), c1 = warp_col(H, qx1);
    const Warped p00 = warp_point(c0, r, qy0), p01 = warp_point(c0, r, qy1);
    const Warped p10 = warp_point(c1, r, qy0), p11 = warp_point(c1, r, qy1);
    const float lo = __builtin_fminf(__builtin_fminf(p00.Z, p01.Z), __builtin_fminf(p10.Z, p11.Z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(p00.Z, p01.Z), __builtin_fmaxf(p10.Z, p11.Z));
    if (!kExactDiv)  // same sign, and magnitudes in [2^-100, 2^100] (NaN fails every comparison)
        return (lo >= 0x1p-100f && hi <= 0x1p100f) || (hi <= -0x1p-100f && lo >= -0x1p100f);
    const bool zok = (lo >= 0x1p-40f && hi <= 0x1p40f) || (hi <= -0x1p-40f && lo >= -0x1p40f);
    const float xm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p00.X), __builtin_fabsf(p01.X)),
                                     __builtin_fmaxf(__builtin_fabsf(p10.X), __builtin_fabsf(p11.X)));
    const float ym = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p00.Y), __builtin_fabsf(p01.Y)),
                                     __builtin_fmaxf(__builtin_fabsf(p10.Y), __builtin_fabsf(p11.Y)));
    // (fmaxf drops a NaN operand: the corner values are tested for NaN through their sum)
    const float s = ((p00.X + p01.X) + (p10.X + p11.X)) + ((p00.Y + p01.Y) + (p10.Y + p11.Y)) + ((p00.Z + p01.Z) + (p10.Z + p11.Z));
    return zok && xm <= 0x1p60f && ym <= 0x1p60f && s == s && __builtin_fabsf(H[2]) >= 0x1p-36f &&
           __builtin_fabsf(H[5]) >= 0x1p-36f;
}
// (the name of rounds 1-5)
__device__ __forceinline__ bool window_z_safe(const float *__restrict__ H, float qx0, float qx1, float qy0, float qy1)
{
    return window_div_safe(H, qx0, qx1, qy0, qy1);
}

// counts significand pairs (x = 1.mx, z = 1.mz) where the Markstein quotient of warp_divide differs from x / z;
// block b takes mz = z_first + b, its lanes all 2^23 mx
__global__ __launch_bounds__(kThreads) void quotient_selftest_kernel(unsigned long long *bad, uint32_t z_first)
{
    const float z = __uint_as_float(0x3f800000u | ((z_first + blockIdx.x) & 0x7fffffu));
    const float r = rcp_correct(z);
    unsigned c = 0;
    for (uint32_t mx = threadIdx.x; mx < (1u << 23); mx += kThreads) {
        const float x = __uint_as_float(0x3f800000u | mx);
        const float q = x * r;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-q, z, x), r, q);
        c += __float_as_uint(q1) != __float_as_uint(x / z);
    }
    if (c) atomicAdd(bad, (unsigned long long)c);
}

// ---------------------------------------------------------------------------------------------
// the 4x4 texel window of a sample (packed 8-bit planes) and the five taps
// ---------------------------------------------------------------------------------------------
// the twelve texels of a 4x4 window the five taps read (corners unused), t<row><col>
struct Tex12 {
    float t01, t02, t10, t11, t12, t13, t20, t21, t22, t23, t31, t32;
};
// window words w0..w3 = columns X..X+3, byte r = row Y+r
__device__ __forceinline__ Tex12 unpack12(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    Tex12 t;
    t.t01 = ub0(w1); t.t02 = ub0(w2);
    t.t10 = ub1(w0); t.t11 = ub1(w1); t.t12 = ub1(w2); t.t13 = ub1(w3);
    t.t20 = ub2(w0); t.t21 = ub2(w1); t.t22 = ub2(w2); t.t23 = ub2(w3);
    t.t31 = ub3(w1); t.t32 = ub3(w2);
    return t;
}

// The coordinate whose floor positions the window and the two values a request keeps for the reduction:
//   model taps (M1)   floor(sx); keeps the fractions (a, b) = (sx - floor sx, sy - floor sy)
//   literal taps      floor(xc) of the CENTRE tap's texture coordinate xc = (sx + 0.5f) - 0.5f (tex2D subtracts the half
//                     texel the caller added, main.cpp:644-648); keeps (sx, sy) -- the five taps form their own fractions
struct WinPos {
    float fx0, fy0;  // floor coordinates of the window's texel (1, 1)
    float ka, kb;    // kept for the reduction
};
__device__ __forceinline__ WinPos win_pos(float sx, float sy)
{
    WinPos p;
    if (kLitTaps) {
        p.fx0 = __builtin_floorf((sx + 0.5f) - 0.5f);
        p.fy0 = __builtin_floorf((sy + 0.5f) - 0.5f);
        p.ka = sx;
        p.kb = sy;
    } else {
        p.fx0 = __builtin_floorf(sx);
        p.fy0 = __builtin_floorf(sy);
        p.ka = sx - p.fx0;
        p.kb = sy - p.fy0;
    }
    return p;
}

#if PM_MODEL & 1
// one texel of a window-packed plane by its image coordinates (clamp-to-edge): byte 0 of word V[y + 3][x + 3]
// (WORDS per texel: 1 gray, 3 colour -- word c of a texel = channel c)
template <int WORDS>
__device__ __forceinline__ float packed_texel(gptr_bytes packed, int pw, int rows, int cols, int chan, int x, int y)
{
    const int xc = clampi(x, 0, cols - 1), yc = clampi(y, 0, rows - 1);
    const unsigned char b = *(const __attribute__((address_space(1))) unsigned char *)(packed + 4 * (((yc + 3) * pw + (xc + 3)) * WORDS + chan));
    return (float)b;
}
// tex2D(tex, x, y) of the reference-on-CPU build's texture model (tex2d_literal in pm_core.h) on a packed plane
template <int WORDS>
__device__ __forceinline__ float tex2d_packed(gptr_bytes packed, int pw, int rows, int cols, int chan, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb);
    const float a = xb - fx, b = yb - fy;
    const int ix = (int)__builtin_fminf(__builtin_fmaxf(fx, -2.0f), (float)cols);
    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fy, -2.0f), (float)rows);
    const float t00 = packed_texel<WORDS>(packed, pw, rows, cols, chan, ix, iy);
    const float t10 = packed_texel<WORDS>(packed, pw, rows, cols, chan, ix + 1, iy);
    const float t01 = packed_texel<WORDS>(packed, pw, rows, cols, chan, ix, iy + 1);
    const float t11 = packed_texel<WORDS>(packed, pw, rows, cols, chan, ix + 1, iy + 1);
    const float v0 = __builtin_fmaf(a, t10 - t00, t00), v1 = __builtin_fmaf(a, t11 - t01, t01);
    return __builtin_fmaf(b, v1 - v0, v0);
}
// The five fetches of gipuma.cu:251-253 one by one -- for the (rare) sample whose separately rounded tap coordinates do not
// land on the neighbouring texels of the centre tap's (lit_fractions: ok false).  dx = +1, -1, 0, 0, 0; dy = 0, 0, +1, -1, 0:
// (sx + dx) + 0.5f is the source's `pt.x + 1 + 0.5f`, `pt.x - 1 + 0.5f`, `pt.x + 0.5f` bit for bit.
template <int WORDS>
__device__ __forceinline__ Taps taps_gather(gptr_bytes packed, int pw, int rows, int cols, int chan, float sx, float sy)
{
    float v[5];
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        const float dx = k == 0 ? 1.0f : k == 1 ? -1.0f : 0.0f, dy = k == 2 ? 1.0f : k == 3 ? -1.0f : 0.0f;
        v[k] = tex2d_packed<WORDS>(packed, pw, rows, cols, chan, (sx + dx) + 0.5f, (sy + dy) + 0.5f);
    }
    Taps o;
    o.gx2 = v[0] - v[1];
    o.gy2 = v[2] - v[3];
    o.sc = v[4];
    return o;
}

// The fractions of the five taps and whether all of them read texels of the window positioned by the centre tap:
// tap (sx +- 1): xb = ((sx +- 1) + 0.5f) - 0.5f must have floor(xb) = floor(xc) +- 1, i.e. xb - (floor(xc) +- 1) in [0, 1) --
// which IS its fraction then (the same subtraction).  A non-negative float below 1 has bits below 0x3f800000; a negative
// one, a NaN or an infinity has more: one unsigned compare of the largest of the four.
struct LitFrac {
    float a0, aL, aR, b0, bU, bD;
    bool ok;
};
__device__ __forceinline__ LitFrac lit_fractions(float sx, float sy)
{
    LitFrac f;
    const float xc = (sx + 0.5f) - 0.5f, yc = (sy + 0.5f) - 0.5f;
    const float fx = __builtin_floorf(xc), fy = __builtin_floorf(yc);
    f.a0 = xc - fx;
    f.b0 = yc - fy;
    f.aL = (((sx - 1) + 0.5f) - 0.5f) - (fx - 1.0f);
    f.aR = (((sx + 1) + 0.5f) - 0.5f) - (fx + 1.0f);
    f.bU = (((sy - 1) + 0.5f) - 0.5f) - (fy - 1.0f);
    f.bD = (((sy + 1) + 0.5f) - 0.5f) - (fy + 1.0f);
    const uint32_t m = max(max(__float_as_uint(f.aL), __float_as_uint(f.aR)), max(__float_as_uint(f.bU), __float_as_uint(f.bD)));
    f.ok = m < 0x3f800000u;
#ifdef PM_LITERAL_FORCE_GATHER  // (test builds: every 8th sample takes the one-by-one path whatever its fractions)
    if ((__float_as_uint(sx) & 7u) == 0u) f.ok = false;
#endif
    return f;
}
// the five fetches from the window's twelve texels: tex2D's own lerps, x first, then y (tex2d_literal), shared wherever
// two taps interpolate the same texel pair with the same fraction -- 28 operations, the bits of five separate fetches
__device__ __forceinline__ Taps taps_literal(const LitFrac &f, const Tex12 &t)
{
    const float r0 = __builtin_fmaf(f.a0, t.t02 - t.t01, t.t01);  // row jc - 1, columns ic, ic + 1
    const float r1 = __builtin_fmaf(f.a0, t.t12 - t.t11, t.t11);  // row jc
    const float r2 = __builtin_fmaf(f.a0, t.t22 - t.t21, t.t21);  // row jc + 1
    const float r3 = __builtin_fmaf(f.a0, t.t32 - t.t31, t.t31);  // row jc + 2
    const float C = __builtin_fmaf(f.b0, r2 - r1, r1);
    const float U = __builtin_fmaf(f.bU, r1 - r0, r0);
    const float D = __builtin_fmaf(f.bD, r3 - r2, r2);
    const float l1 = __builtin_fmaf(f.aL, t.t11 - t.t10, t.t10), l2 = __builtin_fmaf(f.aL, t.t21 - t.t20, t.t20);
    const float L = __builtin_fmaf(f.b0, l2 - l1, l1);
    const float q1 = __builtin_fmaf(f.aR, t.t13 - t.t12, t.t12), q2 = __builtin_fmaf(f.aR, t.t23 - t.t22, t.t22);
    const float R = __builtin_fmaf(f.b0, q2 - q1, q1);
    Taps o;
    o.sc = C;
    o.gx2 = R - L;
    o.gy2 = D - U;
    return o;
}
#endif

// what a reduction needs to know about the packed plane of its view for the one-by-one path of the literal taps
struct PlaneRef {
    gptr_bytes packed;   // the plane itself (not the float-encoded base)
    const Problem *P;    // its geometry (pw, rows, cols), read only on that path
};

// The taps of one sample from its window: (ka, kb) as win_pos kept them, `t` the twelve texels of channel `chan`.
// `frac`: the literal fractions, formed once per sample by the caller (shared by the three channels of a colour sample).
#if PM_MODEL & 1
template <int WORDS>
__device__ __forceinline__ Taps sample_taps(const LitFrac &f, float sx, float sy, const Tex12 &t, const PlaneRef &pr, int chan)
{
    if (f.ok) return taps_literal(f, t);
    return taps_gather<WORDS>(pr.packed, pr.P->pw, pr.P->rows, pr.P->cols, chan, sx, sy);
}
#endif
// gray
__device__ __forceinline__ Taps sample_taps_gray(float ka, float kb, const Tex12 &t, const PlaneRef &pr)
{
#if PM_MODEL & 1
    return sample_taps<1>(lit_fractions(ka, kb), ka, kb, t, pr, 0);
#else
    (void)pr;
    return taps12(ka, kb, t.t01, t.t02, t.t10, t.t11, t.t12, t.t13, t.t20, t.t21, t.t22, t.t23, t.t31, t.t32);
#endif
}
// colour: the three channels' windows
__device__ __forceinline__ void sample_taps_c4(float ka, float kb, const Tex12 &tb, const Tex12 &tg, const Tex12 &tr,
                                               const PlaneRef &pr, Taps (&o)[3])
{
#if PM_MODEL & 1
    const LitFrac f = lit_fractions(ka, kb);
    o[0] = sample_taps<3>(f, ka, kb, tb, pr, 0);
    o[1] = sample_taps<3>(f, ka, kb, tg, pr, 1);
    o[2] = sample_taps<3>(f, ka, kb, tr, pr, 2);
#else
    (void)pr;
    o[0] = taps12(ka, kb, tb.t01, tb.t02, tb.t10, tb.t11, tb.t12, tb.t13, tb.t20, tb.t21, tb.t22, tb.t23, tb.t31, tb.t32);
    o[1] = taps12(ka, kb, tg.t01, tg.t02, tg.t10, tg.t11, tg.t12, tg.t13, tg.t20, tg.t21, tg.t22, tg.t23, tg.t31, tg.t32);
    o[2] = taps12(ka, kb, tr.t01, tr.t02, tr.t10, tr.t11, tr.t12, tr.t13, tr.t20, tr.t21, tr.t22, tr.t23, tr.t31, tr.t32);
#endif
}

// ---------------------------------------------------------------------------------------------
// dis and the accumulation (pmCostComputation_shared, gipuma.cu:263-274; pmCost_shared :672)
// ---------------------------------------------------------------------------------------------
// dis_fold.  gradDis = min((|dgx| + |dgy|) * 0.0625, tau_g) and dis = (1 - alpha) colDis + alpha gradDis.  Scaling by a
// power of two is exact, so with s = |dgx| + |dgy|
//     min(s / 16, tau_g) = min(s, 16 tau_g) / 16      and      alpha * (m / 16) = (alpha / 16) * m
// (the same real number is rounded once, fused or not): the specialised loops take alpha / 16 and 16 tau_g as their
// constants and save the multiplication.  Exact unless alpha / 16 is subnormal or 16 tau_g overflows -- the host checks and
// falls back to the literal loop (view_cost_loop, BOX == 0) -- or s / 16 itself is subnormal (s < 2^-122: a gradient
// difference that small needs a sample within 2^-97 pixels of x = 0; the literal form rounds it, this one does not).
//   gsum = |gradX| + |gradY| (gray) or l1 + l1 (colour), cabs = |colDiff| >= 0 or colDiff itself (ABS: take |.| in the min)
template <bool ABS>
__device__ __forceinline__ float dis_folded(float gsum, float cdiff, float alpha16, float oma, float tau_color, float taug16)
{
    const float gradDis = min_nc(gsum, taug16);
    const float colDis = ABS ? min_abs_nc(cdiff, tau_color) : min_nc(cdiff, tau_color);
    if (kUnfused) return oma * colDis + alpha16 * gradDis;  // (1 - alpha) * colDis + alpha * gradDis, gipuma.cu:272
    return __builtin_fmaf(alpha16, gradDis, oma * colDis);
}
// cost = cost + w * dis, gipuma.cu:274, 672
__device__ __forceinline__ float accum(float w, float dis, float cost)
{
    if (kUnfused) return cost + w * dis;
    return __builtin_fmaf(w, dis, cost);
}
// the product of a chain term where the weight and dis are known before the chain runs (column-per-lane relay): with the
// unfused model the chain adds this product; with the fused one it needs both factors
__device__ __forceinline__ float accum_term(float w, float dis) { return kUnfused ? w * dis : dis; }
__device__ __forceinline__ float accum_add(float w, float term, float cost)
{
    if (kUnfused) return cost + term;
    return __builtin_fmaf(w, term, cost);
}

}  // namespace pm
