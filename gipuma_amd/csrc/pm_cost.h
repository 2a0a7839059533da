// pm_cost.h -- patch costs: pmCost_shared / pmCostComputation_shared (gipuma.cu:585-680, 223-277) as the generic
// loop, the hand-pipelined loop for packed gray planes and its column range variant, the colour loop, the
// column-per-lane evaluation; pmCostMultiview_cu (:720-806) with the register combiner and the exact early
// termination.  Part of the device code of the PatchMatch path (pm_device.h).
#pragma once
#include "pm_core.h"
#include "pm_sample.h"

namespace pm {

// Patch cost of one source view: pmCost_shared + pmCostComputation_shared,
// gipuma.cu:585-680 and :223-277.  `tp0` points at the pixel's own texel inside the LDS tile.
// The arithmetic of a sample -- warp, division, taps, dis, accumulation -- is pm_sample.h's, whatever the model.

// Exponent trick used by the U8 loop: for an integer n in [0, 2^21), the float 2^21 + n has ulp 1/4,
// so its bit pattern is 0x4a000000 + 4n -- a byte offset of 4-byte entry n, produced by a full-rate
// fp32 add instead of cvt + shift (conversions, integer min/max and shifts issue at ~60 % of the
// fp32 rate on gfx950, scripts/ubench/valu_rates.hip).  The constant part moves into the base.
constexpr uint32_t kMagicBits = 0x4a000000u;  // bits of 2^21
constexpr float kMagicF = 0x1p21f;
constexpr int kMagicMaxWords = (1 << 21) - 8;

// ---- shared pieces of the packed-gray loops (float-encoded window offsets) ----
struct MagicAddr {  // wave-uniform: offset of window (Xc, Yc), Xc in [-2, cols], Yc in [-2, rows] (clamped floor coordinates;
    float colsf, rowsf, pwf, magic_c;  // entry (Yc+2)*pw + Xc+2 of V), as the bits of fma(Yc, pw, Xc + magic_c)
#ifdef PM_CHECKED
    const Problem *P;
    uint32_t span;  // bytes of a packed plane a 16-byte window may start in: ((rows + 3) pw) words - 16 bytes + 1
#endif
};
// (-DPM_CHECKED: the float-encoded offset of a window load against the extent of the packed plane)
__device__ __forceinline__ uint32_t magic_checked(const MagicAddr &A, uint32_t off)
{
#ifdef PM_CHECKED
    return kMagicBits + PM_AT(A.P, off - kMagicBits, A.span, kChkWindow);
#else
    (void)A;
    return off;
#endif
}
__device__ __forceinline__ MagicAddr magic_addr(const Problem *__restrict__ P)
{
    MagicAddr A;
    A.colsf = (float)P->cols;
    A.rowsf = (float)P->rows;
    A.pwf = (float)P->pw;
    A.magic_c = kMagicF + (float)(2 * P->pw + 2);
#ifdef PM_CHECKED
    A.P = P;
    A.span = (uint32_t)((P->rows + 3) * P->pw) * 4u - 15u;
#endif
    return A;
}
__device__ __forceinline__ gptr_bytes magic_base_of(const ViewCam &vc)
{
    return (gptr_bytes)((uintptr_t)vc.packed.raw - (uintptr_t)kMagicBits);
}
// (the plane behind a float-encoded base: only the literal taps' one-by-one path reads through it)
__device__ __forceinline__ PlaneRef plane_of_magic(gptr_bytes magic_base, const Problem *__restrict__ P)
{
    PlaneRef pr;
    pr.packed = (gptr_bytes)((uintptr_t)magic_base + (uintptr_t)kMagicBits);
    pr.P = P;
    return pr;
}
struct WinReq {
    float a, b;  // what win_pos keeps: the fractions (model taps) or the sample position itself (literal taps)
    u32x4_a4 w;
};
// the window of the sample (column terms wc, row qy): getCorrespondingPoint_cu, gipuma.cu:207-217, then ONE 16-byte load
template <bool FAST>
__device__ __forceinline__ WinReq magic_request(const MagicAddr &A, gptr_bytes magic_base, const WarpCol &wc,
                                                const WarpRow &wr, float qy)
{
    const Warped p = warp_point(wc, wr, qy);
    float sx, sy;
    warp_divide<FAST>(p, sx, sy);
    const WinPos wp = win_pos(sx, sy);
    WinReq r;
    r.a = wp.ka;
    r.b = wp.kb;
    // v_med3_f32 returns min3 when an input is NaN: NaN -> -2, like the saturating cvt
    const float Xc = __builtin_amdgcn_fmed3f(wp.fx0, -2.0f, A.colsf);
    const float Yc = __builtin_amdgcn_fmed3f(wp.fy0, -2.0f, A.rowsf);
    const uint32_t off = magic_checked(A, __float_as_uint(__builtin_fmaf(Yc, A.pwf, Xc + A.magic_c)));
    r.w = *(gptr_u32x4)(magic_base + off);
    return r;
}
// (FAST as a wave-uniform run-time flag: see GroupWalk::request)
__device__ __forceinline__ WinReq magic_request_rt(const MagicAddr &A, gptr_bytes magic_base, const WarpCol &wc,
                                                   const WarpRow &wr, float qy, bool fast)
{
    const Warped p = warp_point(wc, wr, qy);
    float sx, sy;
#if PM_APPROX
    warp_divide<true>(p, sx, sy);
    (void)fast;
#else
    if (fast)
        warp_divide<true>(p, sx, sy);
    else
        warp_divide<false>(p, sx, sy);
#endif
    const WinPos wp = win_pos(sx, sy);
    WinReq r;
    r.a = wp.ka;
    r.b = wp.kb;
    const float Xc = __builtin_amdgcn_fmed3f(wp.fx0, -2.0f, A.colsf);
    const float Yc = __builtin_amdgcn_fmed3f(wp.fy0, -2.0f, A.rowsf);
    const uint32_t off = magic_checked(A, __float_as_uint(__builtin_fmaf(Yc, A.pwf, Xc + A.magic_c)));
    r.w = *(gptr_u32x4)(magic_base + off);
    return r;
}
struct DisConst {  // wave-uniform: alpha / 16, 1 - alpha, tau_color, 16 tau_gradient (dis_fold, pm_sample.h)
    float alpha16, oma, tau_color, taug16;
};
__device__ __forceinline__ DisConst dis_const(const Problem *__restrict__ P)
{
    DisConst K;
    K.alpha16 = P->alpha * 0.0625f;
    K.oma = 1.f - P->alpha;
    K.tau_color = P->tau_color;
    K.taug16 = P->tau_gradient * 16.0f;
    return K;
}
// dis of a gray sample from its window: the taps, pmCostComputation_shared (gipuma.cu:251-274); (I, gx1, gy1): the reference
// texel of the sample and its central differences
__device__ __forceinline__ float gray_dis(const DisConst &K, const WinReq &cur, float I, float gx1, float gy1, const PlaneRef &pr)
{
    const Tex12 t = unpack12(cur.w.x, cur.w.y, cur.w.z, cur.w.w);
    const Taps tp5 = sample_taps_gray(cur.a, cur.b, t, pr);
    const float colDiff = I - tp5.sc;  // |.| taken in the min
    const float gradX = gx1 - tp5.gx2;
    const float gradY = gy1 - tp5.gy2;
    return dis_folded<true>(__builtin_fabsf(gradX) + __builtin_fabsf(gradY), colDiff, K.alpha16, K.oma, K.tau_color, K.taug16);
}
// support weight of a gray sample from the 256-entry table (weight_cu, gipuma.cu:186-193; images integer valued)
__device__ __forceinline__ float lut_weight(const char *lut_magic, float I, float centre)
{
    const float colorDis = __builtin_fabsf(I - centre);
    return *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
}

template <int BOX, bool U8, bool INTERIOR, bool FAST, bool MAGIC>
__device__ __forceinline__ float view_cost_loop(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py,
                                                const Win<BOX> &win)
{
    const gptr_f32 img = (gptr_f32)vc.img.raw;
    const uint32_t *__restrict__ packed = vc.packed;
    const uint32_t pw = (uint32_t)P->pw;
    const uint32_t xmax = (uint32_t)(P->cols + 2), ymax = (uint32_t)(P->rows + 2);
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const float colsf = (float)cols, rowsf = (float)rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient, gamma = P->gamma;
    const float centre = tp0[0];
    const int hr = win.hrad(), vr = win.vrad();
    const MagicAddr MA = magic_addr(P);
    const gptr_bytes magic_base = magic_base_of(vc);
    const char *lut_magic = (const char *)lut - kMagicBits;
    PlaneRef pr;
    pr.packed = (gptr_bytes)vc.packed.raw;
    pr.P = P;
    float cost = 0.0f;
#if PM_MODEL & 1
    if constexpr (!U8) {
        // Float planes, the reference's own operation order (pmCost_shared, gipuma.cu:633-676; getCorrespondingPoint_cu
        // :207-217; pmCostComputation_shared :251-274) with one gather of four texels per tap: what the CPU restatement
        // computes in its literal flavour 7 and the reference's code compiled for the CPU computes.  -ffp-contract=off keeps
        // every multiply-add unfused.
        for (int i = -hr; i <= hr; i += 2) {
            for (int j = -vr; j <= vr; j += 2) {
                const float4 t4 = *reinterpret_cast<const float4 *>(tp0 + 4 * (j * tw + i));
                const float leftValue = t4.x;
                const float colorDis = __builtin_fabsf(leftValue - centre);
                const float w = exp_model(-colorDis / gamma);  // weight_cu, :186-193
                const float qxf = (float)(px + i), qyf = (float)(py + j);
                // matvecmul4noz, config.h:150-162, then vecdiv4, :44-47
                const float X = H[0] * qxf + H[1] * qyf + H[2];
                const float Y = H[3] * qxf + H[4] * qyf + H[5];
                const float Z = H[6] * qxf + H[7] * qyf + H[8];
                const float sx = X / Z, sy = Y / Z;
                // gipuma.cu:251-253, the argument expressions as written
                const float gx2 = tex2d_literal(img, rows, cols, pitch, sx + 1 + 0.5f, sy + 0.5f) -
                                  tex2d_literal(img, rows, cols, pitch, sx - 1 + 0.5f, sy + 0.5f);
                const float gy2 = tex2d_literal(img, rows, cols, pitch, sx + 0.5f, sy + 1 + 0.5f) -
                                  tex2d_literal(img, rows, cols, pitch, sx + 0.5f, sy - 1 + 0.5f);
                const float colDiff = __builtin_fabsf(leftValue - tex2d_literal(img, rows, cols, pitch, sx + 0.5f, sy + 0.5f));
                const float gradX = t4.y - gx2;  // (right - left) - gx2: the tile holds the central differences
                const float gradY = t4.z - gy2;
                const float gradDis = __builtin_fminf((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
                const float colDis = __builtin_fminf(colDiff, tau_color);
                const float dis = (1.f - alpha) * colDis + alpha * gradDis;  // :272
                cost = cost + w * dis;                                      // :274, :672
            }
        }
        return cost;
    }
#endif
    const WarpRow wr = warp_row(H);
    // (float)(px + i) == (float)px + (float)i exactly (small integers): full-rate adds, no cvt
    float qx = (float)(px - hr);
    for (int i = -hr; i <= hr; i += 2, qx += 2.0f) {
        const WarpCol wc = warp_col(H, qx);
        float qy = (float)(py - vr);
#pragma unroll unroll_j<BOX>()
        for (int j = -vr; j <= vr; j += 2, qy += 2.0f) {
            // one ds_read_b128: {I(q), gx1(q), gy1(q)} of the reference tile
            const float4 t4 = *reinterpret_cast<const float4 *>(tp0 + 4 * (j * tw + i));
            // weight_cu, gipuma.cu:186-193
            float w;
            if (U8)  // images are integer valued in [0,255]: 256 possible weights
                w = lut_weight(lut_magic, t4.x, centre);
            else
                w = exp_model(-__builtin_fabsf(t4.x - centre) / gamma);
            Taps tp5;
            if (U8 && MAGIC) {  // U8 mode: the whole window is one 16-byte load
                const WinReq r = magic_request<FAST>(MA, magic_base, wc, wr, qy);
                tp5 = sample_taps_gray(r.a, r.b, unpack12(r.w.x, r.w.y, r.w.z, r.w.w), pr);
            } else {
                // getCorrespondingPoint_cu, gipuma.cu:207-217
                const Warped p = warp_point(wc, wr, qy);
                float sx, sy;
                warp_divide<FAST>(p, sx, sy);
                const WinPos wp = win_pos(sx, sy);
                if (U8) {
                    // X = clamp(floor(sx), -2, cols) + 2, same for Y: the +2 is exact wherever the
                    // clamp does not saturate
                    const uint32_t X = min(cvt_u32_sat(wp.fx0 + 2.0f), xmax);
                    const uint32_t Y = min(cvt_u32_sat(wp.fy0 + 2.0f), ymax);
                    const uint32_t off = PM_AT(P, (Y * pw + X) << 2, (uint32_t)((P->rows + 3) * P->pw) * 4u - 15u, kChkWindowInt);
                    const u32x4_a4 wv = *(gptr_u32x4)((gptr_bytes)packed + off);
                    tp5 = sample_taps_gray(wp.ka, wp.kb, unpack12(wv.x, wv.y, wv.z, wv.w), pr);
                } else {
                    // float planes (model taps; the literal ones: the branch above): keep the float->int conversion
                    // defined for huge / NaN coordinates
                    const float a = wp.ka, b = wp.kb;
                    const int ix = (int)__builtin_fminf(__builtin_fmaxf(wp.fx0, -2.0f), colsf);
                    const int iy = (int)__builtin_fminf(__builtin_fmaxf(wp.fy0, -2.0f), rowsf);
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
            }
            // pmCostComputation_shared, gipuma.cu:251-274 (the unfolded constants: this is the loop of last resort)
            const float colDiff = t4.w - tp5.sc;  // t4.w == t4.x == I(q); |.| taken in the min
            const float gradX = t4.y - tp5.gx2;
            const float gradY = t4.z - tp5.gy2;
            const float dis = dis_folded<true>((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, colDiff, alpha, oma,
                                               tau_color, tau_gradient);
            cost = accum(w, dis, cost);
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
template <int BOX, bool FAST, bool ET>
__device__ __forceinline__ float view_cost_pipe(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py, float tau,
                                                int *cols_done = nullptr)
{
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2, N = R + 1;  // offsets -R, -R+2, ..., R
    const MagicAddr MA = magic_addr(P);
    const DisConst K = dis_const(P);
    const float centre = tp0[0];
    const gptr_bytes magic_base = magic_base_of(vc);
    const PlaneRef pr = plane_of_magic(magic_base, P);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const WarpRow wr = warp_row(H);

    const float qy0 = (float)(py - R);
    float qx = (float)(px - R);
    WarpCol wc = warp_col(H, qx);
    WinReq r0 = magic_request<FAST>(MA, magic_base, wc, wr, qy0), r1 = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + 2.0f);
    float cost = 0.0f;
    const float *tcol = tp0 + 4 * (-R * tw - R);  // texel (-R, -R) of the window
    for (int c = 0; c < N; c++, tcol += 8) {
        const float qxn = qx + 2.0f;
        const WarpCol wcn = warp_col(H, qxn);
#pragma unroll
        for (int k = 0; k < N; k++) {
            const WinReq cur = r0;
            r0 = r1;
            // sample k+2 of this column, or the first two of the next one
            if (k + 2 < N)
                r1 = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + (float)(2 * (k + 2)));
            else
                r1 = magic_request<FAST>(MA, magic_base, wcn, wr, qy0 + (float)(2 * (k + 2 - N)));
            __builtin_amdgcn_sched_barrier(0);  // keep the request ahead of this sample's reduction
            // one ds_read_b128: {I(q), gx1(q), gy1(q), I(q)} of the reference tile
            const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
            const float w = lut_weight(lut_magic, t4.x, centre);
            const float dis = gray_dis(K, cur, t4.w, t4.y, t4.z, pr);
            cost = accum(w, dis, cost);
            __builtin_amdgcn_sched_barrier(0);
        }
        qx = qxn;
        wc = wcn;
        // early termination (ET): the partial sum only grows (w, dis >= 0, every rounding is monotone),
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
    const MagicAddr MA = magic_addr(P);
    const DisConst K = dis_const(P);
    const float centre = tp0[0];
    const PlaneRef pr = plane_of_magic(magic_base, P);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const WarpRow wr = warp_row(H);

    const float qy0 = (float)(py - R);
    float qx = (float)(px - R + 2 * c0);  // (exact: small integers)
    WarpCol wc = warp_col(H, qx);
    WinReq r0 = magic_request<FAST>(MA, magic_base, wc, wr, qy0), r1 = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + 2.0f);
    const float *tcol = tp0 + 4 * (-R * tw - R) + 8 * c0;
    int c = c0;
    for (; c < c1; c++, tcol += 8) {
        const float qxn = qx + 2.0f;
        const WarpCol wcn = warp_col(H, qxn);
#pragma unroll
        for (int k = 0; k < N; k++) {
            const WinReq cur = r0;
            r0 = r1;
            if (k + 2 < N)
                r1 = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + (float)(2 * (k + 2)));
            else
                r1 = magic_request<FAST>(MA, magic_base, wcn, wr, qy0 + (float)(2 * (k + 2 - N)));
            __builtin_amdgcn_sched_barrier(0);
            const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
            const float w = lut_weight(lut_magic, t4.x, centre);
            const float dis = gray_dis(K, cur, t4.w, t4.y, t4.z, pr);
            cost = accum(w, dis, cost);
            __builtin_amdgcn_sched_barrier(0);
        }
        qx = qxn;
        wc = wcn;
        if (__all(cost >= tau)) {
            c++;
            break;
        }
    }
    if (cols_run) *cols_run += c - c0;
    return cost;
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

// ---- shared pieces of the packed-colour loops (three words per texel: integer window addressing) ----
struct WinReq3 {
    float a, b;  // what win_pos keeps (see WinReq)
    u32x4_a4 q0, q1, q2;
};
struct IntAddr {
    uint32_t pw, xmax, ymax;
#ifdef PM_CHECKED
    const Problem *P;
#endif
};
__device__ __forceinline__ IntAddr int_addr(const Problem *__restrict__ P)
{
    IntAddr A;
    A.pw = (uint32_t)P->pw;
    A.xmax = (uint32_t)(P->cols + 2);
    A.ymax = (uint32_t)(P->rows + 2);
#ifdef PM_CHECKED
    A.P = P;
#endif
    return A;
}
__device__ __forceinline__ WinReq3 c4_window_load(const IntAddr &A, gptr_bytes packed, float sx, float sy)
{
    const WinPos wp = win_pos(sx, sy);
    WinReq3 r;
    r.a = wp.ka;
    r.b = wp.kb;
    // X = clamp(floor, -2, cols) + 2, same for Y: the +2 is exact wherever the clamp does not saturate
    const uint32_t Xw = min(cvt_u32_sat(wp.fx0 + 2.0f), A.xmax);
    const uint32_t Yw = min(cvt_u32_sat(wp.fy0 + 2.0f), A.ymax);
#ifdef PM_CHECKED  // (three words per texel: 48 bytes per window, the plane has (rows + 3) pw texels)
    const gptr_bytes base = packed + PM_AT(A.P, (Yw * A.pw + Xw) * 12u, (uint32_t)((A.P->rows + 3) * A.P->pw) * 12u - 47u, kChkWindowC4);
#else
    const gptr_bytes base = packed + (Yw * A.pw + Xw) * 12u;
#endif
    r.q0 = *(gptr_u32x4)(base);
    r.q1 = *(gptr_u32x4)(base + 16);
    r.q2 = *(gptr_u32x4)(base + 32);
    return r;
}
template <bool FAST>
__device__ __forceinline__ WinReq3 c4_request(const IntAddr &A, gptr_bytes packed, const WarpCol &wc, const WarpRow &wr, float qy)
{
    const Warped p = warp_point(wc, wr, qy);
    float sx, sy;
    warp_divide<FAST>(p, sx, sy);
    return c4_window_load(A, packed, sx, sy);
}
__device__ __forceinline__ WinReq3 c4_request_rt(const IntAddr &A, gptr_bytes packed, const WarpCol &wc, const WarpRow &wr, float qy,
                                                 bool fast)
{
    const Warped p = warp_point(wc, wr, qy);
    float sx, sy;
#if PM_APPROX
    warp_divide<true>(p, sx, sy);
    (void)fast;
#else
    if (fast)
        warp_divide<true>(p, sx, sy);
    else
        warp_divide<false>(p, sx, sy);
#endif
    return c4_window_load(A, packed, sx, sy);
}
// the taps of the three channels of a colour sample: word 3k+c of the 12-word window = column k, channel c
__device__ __forceinline__ void c4_taps(const WinReq3 &r, const PlaneRef &pr, Taps (&t)[3])
{
    const Tex12 tb = unpack12(r.q0.x, r.q0.w, r.q1.z, r.q2.y);
    const Tex12 tg = unpack12(r.q0.y, r.q1.x, r.q1.w, r.q2.z);
    const Tex12 tr = unpack12(r.q0.z, r.q1.y, r.q2.x, r.q2.w);
    sample_taps_c4(r.a, r.b, tb, tg, tr, pr, t);
}
__device__ __forceinline__ PlaneRef plane_of(gptr_bytes packed, const Problem *__restrict__ P)
{
    PlaneRef pr;
    pr.packed = packed;
    pr.P = P;
    return pr;
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
    const gptr_f32 img = (gptr_f32)vc.img.raw;
    const gptr_bytes packed = (gptr_bytes)vc.packed.raw;
    const IntAddr IA = int_addr(P);
    const PlaneRef pr = plane_of(packed, P);
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const float colsf = (float)cols, rowsf = (float)rows;
    // (compile-time boxes: the 1/16 of the gradient term folded into the constants, see dis_fold; BOX == 0 is the
    //  literal fallback the host selects when the folding would not be exact)
    constexpr bool kFold = BOX > 0;
    const float alpha16 = kFold ? P->alpha * 0.0625f : P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, taug16 = kFold ? P->tau_gradient * 16.0f : P->tau_gradient, gamma = P->gamma;
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    const int hr = win.hrad(), vr = win.vrad();
    float cost = cost0;
#if PM_MODEL & 1
    if constexpr (!U8) {
        // Float planes, T = float4 in the reference's own operation order (see view_cost_loop): per channel one bilinear fetch
        // per tap, l1_norm(float4) = (|x| + |y| + |z|) * 0.3333333f (gipuma.cu:174-179), the float4 operators of
        // vector_operations.h.  Whole windows only: the ranged callers (tp_item) run on packed planes.
        (void)c1; (void)cols_run; (void)tau;
        if (c0 != 0 || cost0 != 0.0f) __builtin_trap();
        const float alpha = P->alpha;
        for (int i = -hr; i <= hr; i += 2) {
            for (int j = -vr; j <= vr; j += 2) {
                const float *tp = tp0 + 4 * (j * tw + i);
                const float4 lv = *reinterpret_cast<const float4 *>(tp);
                const float w = exp_model(-l1_3(lv.x - centre.x, lv.y - centre.y, lv.z - centre.z) / gamma);
                const float qxf = (float)(px + i), qyf = (float)(py + j);
                const float X = H[0] * qxf + H[1] * qyf + H[2];
                const float Y = H[3] * qxf + H[4] * qyf + H[5];
                const float Z = H[6] * qxf + H[7] * qyf + H[8];
                const float sx = X / Z, sy = Y / Z;
                const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
                const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
                const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
                const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
                const float lvc[3] = {lv.x, lv.y, lv.z};
                const float gx1[3] = {right.x - left.x, right.y - left.y, right.z - left.z};
                const float gy1[3] = {down.x - up.x, down.y - up.y, down.z - up.z};
                float cd[3], gX[3], gY[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float gx2 = tex2d_literal<4>(img + c, rows, cols, pitch, sx + 1 + 0.5f, sy + 0.5f) -
                                      tex2d_literal<4>(img + c, rows, cols, pitch, sx - 1 + 0.5f, sy + 0.5f);
                    const float gy2 = tex2d_literal<4>(img + c, rows, cols, pitch, sx + 0.5f, sy + 1 + 0.5f) -
                                      tex2d_literal<4>(img + c, rows, cols, pitch, sx + 0.5f, sy - 1 + 0.5f);
                    cd[c] = lvc[c] - tex2d_literal<4>(img + c, rows, cols, pitch, sx + 0.5f, sy + 0.5f);
                    gX[c] = gx1[c] - gx2;
                    gY[c] = gy1[c] - gy2;
                }
                const float colDiff = l1_3(cd[0], cd[1], cd[2]);
                const float gradDis = __builtin_fminf((l1_3(gX[0], gX[1], gX[2]) + l1_3(gY[0], gY[1], gY[2])) * 0.0625f, P->tau_gradient);
                const float colDis = __builtin_fminf(colDiff, tau_color);
                const float dis = (1.f - alpha) * colDis + alpha * gradDis;
                cost = cost + w * dis;
            }
        }
        return cost;
    }
#endif
    const WarpRow wr = warp_row(H);
    float qx = (float)(px - hr + 2 * c0);
    int col = c0;
    for (int i = -hr + 2 * c0; i <= hr && col < c1; i += 2, qx += 2.0f) {
        const WarpCol wc = warp_col(H, qx);
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
            Taps t[3];
            if (U8) {
                c4_taps(c4_request<FAST>(IA, packed, wc, wr, qy), pr, t);
            } else {
                const Warped p = warp_point(wc, wr, qy);
                float sx, sy;
                warp_divide<FAST>(p, sx, sy);
                const WinPos wp = win_pos(sx, sy);  // (model taps; the literal ones on float planes: the branch above)
                const float a = wp.ka, b = wp.kb;
                const int ix = (int)__builtin_fminf(__builtin_fmaxf(wp.fx0, -2.0f), colsf);
                const int iy = (int)__builtin_fminf(__builtin_fmaxf(wp.fy0, -2.0f), rowsf);
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
            const float dis = dis_folded<false>(kFold ? gX + gY : (gX + gY) * 0.0625f, colDiff, alpha16, oma, tau_color, taug16);
            cost = accum(w, dis, cost);
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
// reference's summation order (columns outer, rows inner, one accumulation per sample into a single
// accumulator, gipuma.cu:633-676) is kept by a relay: each lane stores the N terms of its
// column; in relay step c every lane re-runs its N accumulations starting from the value its left
// neighbour produced in step c-1, so after step c lane c holds the exact prefix over columns 0..c
// (the other lanes' values are never used).  N*(N+1) extra instructions per N samples per lane --
// irrelevant where the launch is bound by line fills.
// ---------------------------------------------------------------------------------------------
#ifndef PM_COLS_PD
#define PM_COLS_PD 4  // (round 6, planes ordered by disparity: 8 -> 4 requests in flight is level to -1 % per view, 10 registers fewer)
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

#if PM_APPROX && defined(PM_APPROX_TREE_SUM)
// (A/B builds) instead of the relay below: every lane sums its own window column, the columns of a group are added by a
// butterfly over the group's lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): log2(G) DPP adds instead of
// N relay steps of N fmaf each; every lane of the group ends with the total
template <int BOX>
__device__ __forceinline__ float cols_tree_sum(const float *wgt, const float *dis, int col)
{
    constexpr int N = (BOX + 1) / 2, G = col_group<BOX>();
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < N; k++) acc = __builtin_fmaf(wgt[k], dis[k], acc);
    if (col >= N) acc = 0.0f;  // spare lanes shadow the last column
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x141, 0xf, 0xf, false));  // row_half_mirror
    if (G == 16) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x140, 0xf, 0xf, false));  // row_mirror
    return acc;
}
#endif

template <int BOX, bool FAST>
__device__ __forceinline__ float view_cost_cols(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tw, const float *__restrict__ lut, int px, int py, int col)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(BOX > 0 && N <= col_group<BOX>(), "one lane per window column");
    const MagicAddr MA = magic_addr(P);
    const DisConst K = dis_const(P);
    const float centre = tp0[0];
    const gptr_bytes magic_base = magic_base_of(vc);
    const PlaneRef pr = plane_of_magic(magic_base, P);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const WarpRow wr = warp_row(H);
    const int mycol = col < N ? col : N - 1;  // spare lanes of a smaller box shadow the last column
    const float qx = (float)(px - R + 2 * mycol);
    const WarpCol wc = warp_col(H, qx);

    const float qy0 = (float)(py - R);
    const float *tcol = tp0 + 4 * (-R * tw - R + 2 * mycol);  // texel (column, -R) of the window
    // per row of the lane's column: the weight and dis (fused model) or their product (unfused: what the chain adds)
    float wgt[N], dis[N];
    // PD window requests in flight (these launches wait on L2 misses, and the kernel has registers
    // to spare below its 3-wavefront budget)
    constexpr int PDmax = N > 8 ? 4 : PM_COLS_PD;  // (13 samples per column: keep the registers for wgt/dis)
    constexpr int PD = PDmax < N ? PDmax : N;
    WinReq req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + (float)(2 * p));
#pragma unroll
    for (int k = 0; k < N; k++) {
        const WinReq cur = req[k % PD];
        if (k + PD < N) req[k % PD] = magic_request<FAST>(MA, magic_base, wc, wr, qy0 + (float)(2 * (k + PD)));
        const float4 t4 = *reinterpret_cast<const float4 *>(tcol + 8 * k * tw);
        wgt[k] = lut_weight(lut_magic, t4.x, centre);
        dis[k] = accum_term(wgt[k], gray_dis(K, cur, t4.w, t4.y, t4.z, pr));
    }
#if PM_APPROX && defined(PM_APPROX_TREE_SUM)  // (A/B builds only: measured, no gain, costs agreement)
    return cols_tree_sum<BOX>(wgt, dis, col);
#endif
    // relay: after step c, lane c of the group holds the sum over columns 0..c in reference order
    float out = 0.0f;
#pragma unroll
    for (int c = 0; c < N; c++) {
        // lane i takes lane i-1's value: DPP row_shr:1 (groups of 8 never straddle a row of 16)
        float acc = c == 0 ? 0.0f
                           : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out), 0x111, 0xf, 0xf, false));
#pragma unroll
        for (int k = 0; k < N; k++) acc = accum_add(wgt[k], dis[k], acc);
        out = acc;
    }
    return out;  // exact in lane N-1 of the group
}

// view_cost_cols for T = float4 (-color_processing): lane c evaluates window column c by the
// arithmetic of view_cost_c4_loop (three 16-byte window loads and tap sets per sample, l1_norm(float4)
// reductions, weight table indexed by |dB|+|dG|+|dR|, integer window addressing), the relay keeps the
// reference's summation order.  `tp0` points at the pixel's own texel in the float4 {B, G, R, 0} tile.
template <int BOX, bool FAST>
__device__ __forceinline__ float view_cost_cols_c4(const Problem *__restrict__ P, const ViewCam &vc,
                                                   const float *__restrict__ H, const float *__restrict__ tp0,
                                                   int tw, const float *__restrict__ lut, int px, int py, int col)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(BOX > 0 && N <= col_group<BOX>(), "one lane per window column");
    const gptr_bytes packed = (gptr_bytes)vc.packed.raw;
    const IntAddr IA = int_addr(P);
    const PlaneRef pr = plane_of(packed, P);
    const DisConst K = dis_const(P);
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    const WarpRow wr = warp_row(H);
    const int mycol = col < N ? col : N - 1;
    const float qx = (float)(px - R + 2 * mycol);
    const WarpCol wc = warp_col(H, qx);

    const float qy0 = (float)(py - R);
    const float *tcol = tp0 + 4 * (-R * tw - R + 2 * mycol);  // texel (column, -R) of the window
    float wgt[N], dis[N];
#ifndef PM_COLS_C4_PD
#define PM_COLS_C4_PD 3
#endif
    constexpr int PD = PM_COLS_C4_PD < N ? PM_COLS_C4_PD : N;  // window requests (three loads each) in flight
    WinReq3 req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = c4_request<FAST>(IA, packed, wc, wr, qy0 + (float)(2 * p));
#pragma unroll
    for (int k = 0; k < N; k++) {
        const WinReq3 cur = req[k % PD];
        if (k + PD < N) req[k % PD] = c4_request<FAST>(IA, packed, wc, wr, qy0 + (float)(2 * (k + PD)));
        const float *tp = tcol + 8 * k * tw;
        const float4 lv = *reinterpret_cast<const float4 *>(tp);
        const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                        __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
        wgt[k] = lut[(int)S];
        Taps t[3];
        c4_taps(cur, pr, t);
        const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
        const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
        const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
        const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
        const float colDiff = l1_3(lv.x - t[0].sc, lv.y - t[1].sc, lv.z - t[2].sc);
        const float gX = l1_3((right.x - left.x) - t[0].gx2, (right.y - left.y) - t[1].gx2,
                              (right.z - left.z) - t[2].gx2);
        const float gY = l1_3((down.x - up.x) - t[0].gy2, (down.y - up.y) - t[1].gy2,
                              (down.z - up.z) - t[2].gy2);
        dis[k] = accum_term(wgt[k], dis_folded<false>(gX + gY, colDiff, K.alpha16, K.oma, K.tau_color, K.taug16));
    }
#if PM_APPROX && defined(PM_APPROX_TREE_SUM)  // (A/B builds only: measured, no gain, costs agreement)
    return cols_tree_sum<BOX>(wgt, dis, col);
#endif
    // relay: after step c, lane c of the group holds the sum over columns 0..c in reference order
    float out = 0.0f;
#pragma unroll
    for (int c = 0; c < N; c++) {
        float acc = c == 0 ? 0.0f
                           : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out), 0x111, 0xf, 0xf, false));
#pragma unroll
        for (int k = 0; k < N; k++) acc = accum_add(wgt[k], dis[k], acc);
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

}  // namespace pm
