// pm_prefilter.h -- lower-bound prefilter of refinement candidates: the per-pixel list of the heaviest window
// samples (weight_order_kernel) and the bound they give for one (candidate, view) item (lb_item, lb_item_c4).
// Part of the device code of the PatchMatch path (pm_device.h).
#pragma once
#include "pm_core.h"
#include "pm_cost.h"

namespace pm {

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
// Rigour against rounding.  Let C be the reference's chain value (n <= 169 accumulations in window order: box <= 25),
// T the exact real sum of its terms, l the value lb_item accumulates over S of the terms (|S| = k <= 32, any order) and
// T_S <= T their exact sum, u = 2^-24.  Every accumulation rounds non-negative exact values to nearest -- once (fmaf, model
// bit 2 clear) or twice (the product, then the sum: the unfused models; r = 1 or 2 roundings) --, so
// C >= T (1-u)^(r n) - r n 2^-150 and l <= T_S (1+u)^(r k) + r k 2^-150 (the absolute terms cover subnormal partial sums).
// Hence C >= l (1 - r (n + k + 1) u) - 2^-140, and for l >= 2^-60
//     L' = l * (1 - 2^-16)   fused: r (n + k + 1) u <= 202 u < 256 u = 2^-16 (one more rounding included)
//     L' = l * (1 - 2^-15)   unfused: 404 u < 512 u
// satisfies L' <= C.  An item with L' >= thr is decided: its view cost is at least L', which is what
// the ViewCombiner gets -- "a lower bound >= thr", the case multiview_cost's proof calls an abandoned
// view.  Everything else about refine_two_phase is unchanged.
// ---------------------------------------------------------------------------------------------
constexpr float kLbShrink = kUnfused ? 0.999969482421875f : 0.9999847412109375f;  // 1 - 2^-15 : 1 - 2^-16
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
    constexpr int KMAX = lb_max<BOX>();
    constexpr int G = PM_LB_GROUP, NG = N / G, KG = KMAX / G;  // group size, groups per window row, groups listed
    static_assert(G == 1 || G == 2 || G == 4, "group size");
    static_assert(NG * N >= KG, "the window has enough groups");
    const int rows = P->rows, cols = P->cols, pitch = P->pitch;
    const int np = rows * cols;
    const int center = blockIdx.x * kThreads + threadIdx.x;
    if (center >= np) return;
    const int py = center / cols, px = center - py * cols;
    const gptr_f32 ref = (gptr_f32)P->ref.raw;
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
    for (int d = 0; d < KMAX / 2; d++) {
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
                                         const uint32_t *__restrict__ ordp, size_t np,
                                         const uint32_t (&ord)[kLbRegDwords], int kd, float *lb_short)
{
    // (*lb_short: the sum two samples short of the end -- what the probe workgroups use to judge the length)
    // Lists of up to 16 samples (boxes 11, 15) are handed over in registers `ord`, loaded once per refinement step
    // and walked by an unrolled loop; longer ones (box 25) are streamed from `ordp` -- on config C the streamed,
    // rolled loop was 2.5 % slower per view than the unrolled one (90.8 vs 88.5 ms).
    static_assert(BOX > 0, "compile-time window only");
    constexpr int R = (BOX - 1) / 2;
    const MagicAddr MA = magic_addr(P);
    const DisConst K = dis_const(P);
    const PlaneRef pr = plane_of_magic(magic_base, P);
    const float centre = tp0[0];
    const char *lut_magic = (const char *)lut - kMagicBits;
    const WarpRow wr = warp_row(H);
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
        const WinReq q = magic_request<FAST>(MA, magic_base, warp_col(H, qx), wr, qy);
        LbReq r;
        r.a = q.a;
        r.b = q.b;
        r.w = q.w;
        r.taddr = __float_as_uint(__builtin_fmaf(rif, trow, __builtin_fmaf(cif, 32.0f, tbias)));
        return r;
    };
    auto reduce = [&](const LbReq &cur, float acc) -> float {
        const float4 t4 = *reinterpret_cast<const float4 *>(tile_magic + cur.taddr);
        const float w = lut_weight(lut_magic, t4.x, centre);
        WinReq q;
        q.a = cur.a;
        q.b = cur.b;
        q.w = cur.w;
        return accum(w, gray_dis(K, q, t4.w, t4.y, t4.z, pr), acc);
    };

    float lb = 0.0f, prev = 0.0f;
    if constexpr (lb_in_registers<BOX>()) {
        LbReq r0 = request(ub0(ord[0]), ub1(ord[0])), r1 = request(ub2(ord[0]), ub3(ord[0]));
#pragma unroll
        for (int d = 0; d < kLbRegDwords; d++) {
            if (d >= kd) break;  // (wave-uniform)
            prev = lb;
            // (the two requests past the last sample fetch valid, clamped addresses and are dropped)
            const uint32_t cw = ord[d + 1 < kLbRegDwords ? d + 1 : d];
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
    typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
    const gptr_u32 op = (gptr_u32)ordp;
    const uint32_t w0 = op[0];
    uint32_t nxt = op[kd > 1 ? np : 0];
    LbReq r0 = request(ub0(w0), ub1(w0)), r1 = request(ub2(w0), ub3(w0));
    for (int d = 0; d < kd; d++) {
        prev = lb;
        // samples 2d + 2 and 2d + 3 are requested while 2d and 2d + 1 are reduced; the list word after them is
        // on its way (the two requests past the last sample fetch valid, clamped addresses and are dropped)
        const uint32_t cw = nxt;
        nxt = op[(size_t)min(d + 2, lb_max<BOX>() / 2 - 1) * np];
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
    const gptr_bytes packed = (gptr_bytes)vc.packed.raw;
    const IntAddr IA = int_addr(P);
    const PlaneRef pr = plane_of(packed, P);
    const DisConst K = dis_const(P);
    const float4 centre = *reinterpret_cast<const float4 *>(tp0);
    const WarpRow wr = warp_row(H);
    typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
    const gptr_u32 op = (gptr_u32)ordp;
    struct Req3 {
        WinReq3 q;
        const float *tp;
    };
    auto request = [&](uint32_t cw, int e) -> Req3 {
        const int ci = (int)((cw >> (16 * e)) & 255u), ri = (int)((cw >> (16 * e + 8)) & 255u);
        const int i = 2 * ci - R, j = 2 * ri - R;
        const float qx = (float)(px + i), qy = (float)(py + j);
        Req3 r;
        r.q = c4_request<FAST>(IA, packed, warp_col(H, qx), wr, qy);
        r.tp = tp0 + 4 * (j * tw + i);
        return r;
    };
    auto reduce = [&](const Req3 &c, float acc) -> float {
        const float *tp = c.tp;
        const float4 lv = *reinterpret_cast<const float4 *>(tp);
        const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                        __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
        const float w = lut[(int)S];
        Taps t[3];
        c4_taps(c.q, pr, t);
        const float4 up = *reinterpret_cast<const float4 *>(tp - 4 * tw);
        const float4 down = *reinterpret_cast<const float4 *>(tp + 4 * tw);
        const float4 left = *reinterpret_cast<const float4 *>(tp - 4);
        const float4 right = *reinterpret_cast<const float4 *>(tp + 4);
        const float colDiff = l1_3(lv.x - t[0].sc, lv.y - t[1].sc, lv.z - t[2].sc);
        const float gX = l1_3((right.x - left.x) - t[0].gx2, (right.y - left.y) - t[1].gx2,
                              (right.z - left.z) - t[2].gx2);
        const float gY = l1_3((down.x - up.x) - t[0].gy2, (down.y - up.y) - t[1].gy2,
                              (down.z - up.z) - t[2].gy2);
        return accum(w, dis_folded<false>(gX + gY, colDiff, K.alpha16, K.oma, K.tau_color, K.taug16), acc);
    };
    float lb = 0.0f, prev = 0.0f;
    // (requesting sample s + 1 before sample s is reduced was measured level on the colour workload: 231.5 vs 230.9 ms)
    for (int d = 0; d < kd; d++) {
        prev = lb;
        const uint32_t cw = op[(size_t)d * np];
        lb = reduce(request(cw, 0), lb);
        lb = reduce(request(cw, 1), lb);
    }
    *lb_short = prev;
    return lb;
}

}  // namespace pm
