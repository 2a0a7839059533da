// pm_push.h -- spatial propagation turned round: every plane is evaluated ONCE, by its owner, for
// all the pixels it will be offered to.
//
// What it replaces: the cost evaluations inside gipuma_checkerboard_spatialPropClose_cu / ...Far_cu
// (reference gipuma.cu:1471-1588, 1353-1468; pmCostMultiview_cu :720-806 at :865-872) of the NEXT
// half-sweep.  The accept tests themselves stay with the consumer (sweep_replay in pm_device.h).
//
// Observation (exact).  The patch cost of plane pi at pixel p in view v is
//     c_v(p, pi) = sum over the window samples q = p + (2i-R, 2j-R) of  w(p, q) * dis_v(q, pi)
// accumulated by fmaf, i outer, j inner (gipuma.cu:633-676).  dis_v(q, pi) -- the warp of q through
// the homography of (pi, v), the five bilinear taps, the truncated colour / gradient differences
// (gipuma.cu:207-274) -- does not depend on p.  The plane of pixel n (a PRODUCER) is a propagation
// candidate of exactly eight pixels of the other colour, its CONSUMERS n+-1 and n+-5 in x and in y
// (gipuma.cu:1437-1462, 1571-1582), whose windows overlap: the four consumers on n's column need
// 8 x 13 distinct samples instead of 4 x 64 (window offsets are odd, so the rows they touch are
// n.y - 12, -10, ... +12), the four on n's row 13 x 8.  208 sample points instead of 512.
//
// So after a half-sweep of colour X has fixed the planes of X, this kernel (same colour X)
//   eval   evaluates dis_v(q, plane(n)) once on n's 208-point stencil, a group of 8 lanes per
//          producer, 8 consecutive points of a stencil row per step (the lanes of a load share
//          cache lines whatever the plane -- what the column-per-lane kernel is built for), by the
//          instruction sequence of view_cost_pipe, and leaves the values in LDS;
//   chain  lane c of the group then runs the reference's 64-term fmaf chain for consumer c --
//          its support weights w(p_c, q) (weight_cu, gipuma.cu:186-193), the terms in the
//          reference's order -- and feeds the view cost to its ViewCombiner;
// and after the last view stores the aggregate F(p_c, plane(n)) in Problem::push_cost[slot c][p_c],
// where the next half-sweep (colour 1-X) finds it instead of evaluating it (Tune::kPushConsume).
// Same terms, same order, same roundings as view_cost_pipe + multiview_cost: bit-identical.
//
// Rule (H) of sweep_kernel carries over unchanged: when the host knows that the next half-sweep
// may use it (`hist`), a producer whose plane did not change in this half-sweep offers nothing, and
// its consumers skip that slot (they read the same Problem::changed flag).  Rules (A) and (D) are
// not needed: a consumer that replays a cost it would have skipped rejects it (see sweep_kernel).
//
// No workgroup barrier after the set-up: a group's dis values are written and read by lanes of one
// wavefront (LDS operations of a wavefront complete in order).
//
// Supported: gray window-packed planes with float-encoded offsets, box 15, best-N with n_best <= 4.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int kPushReach = 5;                         // propagation distance (gipuma.cu:1437-1462)
constexpr int kPushLanes = 8;                         // lanes per producer
constexpr int kPushGroups = kThreads / kPushLanes;    // producers evaluated concurrently by a workgroup

template <int BOX>
struct PushLayout {  // offsets in 32-bit words into the dynamic LDS array
    static_assert(BOX == 15, "8 window columns = 8 lanes");
    static constexpr int R = (BOX - 1) / 2, N = R + 1;
    static constexpr int FWH = N + kPushReach;        // 13: rows of the vertical family / columns of the horizontal one
    static constexpr int NF = N * FWH;                // 104 points per family
    static constexpr int halo = R + kPushReach + 1;   // 13: samples reach 12 texels, their gradients one more
    static constexpr int tw = kTileW + 2 * halo, th = kSweepTileH + 2 * halo;  // 58 x 42
    // only texels of the consumers' colour are ever sampled (producer + odd + even offsets): the tile
    // is stored checkerboard-compressed, entry (ty, tx >> 1)
    static constexpr int twc = (tw + 1) / 2;          // 29 (odd: rows land in different banks)
    static constexpr int hbase = NF;                  // horizontal family behind the vertical one
    static constexpr int dstride = 2 * NF + 2;        // words per group (bank spread, scripts/exp/push_banks.py)
    static constexpr int tile4 = kLutSize;            // {I, gx1, gy1, I} per compressed texel
    static constexpr int iplane = tile4 + 4 * twc * th;  // I alone (conflict-free 4-byte reads of the chain)
    static constexpr int dis = iplane + twc * th;     // [kPushGroups][dstride]; before that the staging plane
    static constexpr int list = dis + kPushGroups * dstride;  // 256 u16: producers with something to offer
    static constexpr int cnt = list + kThreads / 2;
    static constexpr int total = cnt + 8;
    static_assert(tw * th <= kPushGroups * dstride, "the staging plane aliases the sample buffers");
};

// consumer c of a producer = the pixel that meets the producer as its neighbour slot c
// (pm::neighbour: 0 up, 1 down, 2 left, 3 right at distance 1, 4..7 at distance 5)
__device__ __forceinline__ void push_consumer_offset(int c, int &dx, int &dy)
{
    const int d = c < 4 ? 1 : kPushReach;
    const int k = c & 3;
    dx = k == 2 ? d : k == 3 ? -d : 0;
    dy = k == 0 ? d : k == 1 ? -d : 0;
}

// dis of the 208 stencil points of one (producer, view) pair: 26 steps of 8 lanes.
//   steps 0..12   vertical family, point (x = lane, y = step): offset (2x-7, 2y-12) from the producer
//   steps 13..25  horizontal family, 13 points per row dealt 8 at a time: offset (2x-12, 2y-7)
// (scripts/exp/push_model.py checks these index formulas against the definition)
template <int BOX, bool FAST>
__device__ __forceinline__ void push_eval(const Problem *__restrict__ P, gptr_bytes magic_base,
                                          const float *__restrict__ H, const float *__restrict__ tile4,
                                          int base_v, int base_h, float *__restrict__ dgrp, float nxf, float nyf,
                                          int l)
{
    using LY = PushLayout<BOX>;
    constexpr int FWH = LY::FWH, twc = LY::twc, S = 2 * FWH;
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const float H0 = H[0], H1 = H[1], H2 = H[2], H3 = H[3], H4 = H[4], H5 = H[5], H6 = H[6], H7 = H[7], H8 = H[8];

    auto request = [&](float qx, float qy) -> WinReq {
        // getCorrespondingPoint_cu, gipuma.cu:207-217, the fmaf nesting of view_cost_pipe
        const float X = __builtin_fmaf(H1, qy, __builtin_fmaf(H0, qx, H2));
        const float Y = __builtin_fmaf(H4, qy, __builtin_fmaf(H3, qx, H5));
        const float Z = __builtin_fmaf(H7, qy, __builtin_fmaf(H6, qx, H8));
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
    // (float)(n + d) == (float)n + (float)d exactly (small integers)
    const float qx_v = nxf + (float)(2 * l - LY::R);                    // vertical family: the lane's column
    const float qx_h = nxf + (float)(2 * l - (LY::R + kPushReach));     // horizontal family, before the row wrap
    auto issue = [&](int s) -> WinReq {
        if (s < FWH) return request(qx_v, nyf + (float)(2 * s - (LY::R + kPushReach)));
        const int t = s - FWH;
        const int j0 = (8 * t) / FWH, r0 = (8 * t) % FWH;
        const bool wrap = (FWH - r0 < kPushLanes) && l >= FWH - r0;
        const float qx = qx_h + (wrap ? (float)(2 * r0 - 2 * FWH) : (float)(2 * r0));
        const float qy = nyf + (wrap ? (float)(2 * j0 + 2 - LY::R) : (float)(2 * j0 - LY::R));
        return request(qx, qy);
    };
    auto tile_index = [&](int s) -> int {
        if (s < FWH) return base_v + (2 * s - (LY::R + kPushReach)) * twc;
        const int t = s - FWH;
        const int j0 = (8 * t) / FWH, r0 = (8 * t) % FWH;
        const bool wrap = (FWH - r0 < kPushLanes) && l >= FWH - r0;
        return base_h + (2 * j0 - LY::R) * twc + r0 + (wrap ? 2 * twc - FWH : 0);
    };

    constexpr int PD = 8;  // window requests in flight per lane
    WinReq req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = issue(p);
#pragma unroll
    for (int s = 0; s < S; s++) {
        const WinReq cur = req[s % PD];
        if (s + PD < S) req[s % PD] = issue(s + PD);
        // {I(q), gx1(q), gy1(q), I(q)} of the reference tile
        const float4 t4 = *reinterpret_cast<const float4 *>(tile4 + 4 * tile_index(s));
        const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
        // pmCostComputation_shared, gipuma.cu:251-274
        const float colDiff = t4.w - tp5.sc;
        const float gradX = t4.y - tp5.gx2;
        const float gradY = t4.z - tp5.gy2;
        const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
        const float colDis = min_abs_nc(colDiff, tau_color);
        dgrp[(s < FWH ? 0 : LY::hbase - 8 * FWH) + 8 * s + l] = __builtin_fmaf(alpha, gradDis, oma * colDis);
    }
}

// the reference's summation for one consumer: columns outer, rows inner, one fmaf per sample
// (gipuma.cu:633-676); `ipl` = the compressed I plane at the consumer's window corner, `dch` = the
// group's sample buffer at the consumer's first sample, rows `jstride` apart
template <int BOX>
__device__ __forceinline__ float push_chain(const float *__restrict__ ipl, float centre, const char *lut_magic,
                                            const float *__restrict__ dch, int jstride)
{
    using LY = PushLayout<BOX>;
    float cost = 0.0f;
#pragma unroll
    for (int i = 0; i < LY::N; i++)
#pragma unroll
        for (int j = 0; j < LY::N; j++) {
            // weight_cu, gipuma.cu:186-193: 256 possible weights
            const float colorDis = __builtin_fabsf(ipl[2 * j * LY::twc + i] - centre);
            const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
            cost = __builtin_fmaf(w, dch[j * jstride + i], cost);
        }
    return cost;
}

// grid = the sweep tiles of the frame; `colour` = the colour of the producers (the colour that was
// swept last); hist: offer only the planes that changed in that half-sweep (rule (H))
template <int BOX>
__global__ __launch_bounds__(kThreads, 3) void push_kernel(const Problem *__restrict__ P,
                                                           const float4 *__restrict__ norm4, int colour, int hist,
                                                           unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = PushLayout<BOX>;
    constexpr int R = LY::R, N = LY::N, twc = LY::twc, halo = LY::halo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = P->rows, cols = P->cols;
    const int gx = (cols + kTileW - 1) / kTileW;
    const int gy = (rows + kSweepTileH - 1) / kSweepTileH;
    const TileXY txy = tile_of(blockIdx.x, gx, gy, tune);
    const int x0 = txy.x * kTileW, y0 = txy.y * kSweepTileH;
    float *tile4 = lds + LY::tile4;
    float *iplane = lds + LY::iplane;
    unsigned short *list = reinterpret_cast<unsigned short *>(lds + LY::list);
    int *cnt = reinterpret_cast<int *>(lds + LY::cnt);

    // ---- producers of this tile with something to offer ----
    const int ly = tid >> 4, lx = 2 * (tid & 15) + ((ly + colour) & 1);  // the lane mapping of sweep_read_state
    const int px = x0 + lx, py = y0 + ly;
    const bool inside = px < cols && py < rows;
    const bool live = inside && (!hist || P->changed[py * cols + px] != 0);
    const unsigned long long bal = __ballot(live);
    if (lane == 0) cnt[wave] = (int)__popcll(bal);

    // ---- reference tile (clamp-to-edge point samples like the reference's, gipuma.cu:1393-1402),
    //      checkerboard-compressed, with the gradients of pmCostComputation_shared (:254-259) ----
    {
        constexpr int tw = LY::tw, th = LY::th;
        const gptr_f32 ref = (gptr_f32)P->ref;
        float *plane = lds + LY::dis;
        for (int k = tid; k < tw * th; k += kThreads) {
            const int ty = k / tw, tx = k - ty * tw;
            const int sx = clampi(x0 - halo + tx, 0, cols - 1);
            const int sy = clampi(y0 - halo + ty, 0, rows - 1);
            plane[k] = ref[sy * P->pitch + sx];
        }
        for (int k = tid; k < kLutSize; k += kThreads) lds[k] = exp_model(-(float)k / P->gamma);
        __syncthreads();
        const int cpar = 1 - colour;  // (x + y) & 1 of the consumers and of every sample point
        for (int k = tid; k < twc * th; k += kThreads) {
            const int ty = k / twc, cx = k - ty * twc;
            const int tx = 2 * cx + ((cpar + ty) & 1);  // the tile origin (x0 - 13, y0 - 13) is even + even
            float I = 0.0f, gx1 = 0.0f, gy1 = 0.0f;
            if (tx < tw) {
                I = plane[ty * tw + tx];
                if (tx > 0 && tx < tw - 1 && ty > 0 && ty < th - 1) {
                    gx1 = plane[ty * tw + tx + 1] - plane[ty * tw + tx - 1];
                    gy1 = plane[(ty + 1) * tw + tx] - plane[(ty - 1) * tw + tx];
                }
            }
            *reinterpret_cast<float4 *>(tile4 + 4 * k) = make_float4(I, gx1, gy1, I);
            iplane[k] = I;
        }
    }
    const int c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];  // (written before the first barrier above)
    const int n_live = c0 + c1 + c2 + c3;
    if (live) {
        const int first = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        list[first + rank] = (unsigned short)tid;
    }
    __syncthreads();  // tile, table and list complete; the staging plane is dead
    if (n_live == 0) return;

    const int n = P->n_sel;
    const size_t np = (size_t)rows * (size_t)cols;
    const int grp = tid / kPushLanes, l = tid % kPushLanes;
    const int grp_lane0 = lane & ~(kPushLanes - 1);
    float *dgrp = lds + LY::dis + grp * LY::dstride;
    const char *lut_magic = (const char *)lds - kMagicBits;
    int cdx, cdy;
    push_consumer_offset(l, cdx, cdy);
    // where consumer l's samples start in the group's buffer, and how far its window rows are apart
    const int dbase = cdx == 0 ? ((cdy + kPushReach) / 2) * N : LY::hbase + (cdx + kPushReach) / 2;
    const int jstride = cdx == 0 ? N : LY::FWH;

    for (int r = 0; r * kPushGroups < n_live; r++) {
        const int idx = r * kPushGroups + grp;
        const bool have = idx < n_live;
        if (!__any(have)) break;  // (later wavefronts hold the larger indices)
        const int ptid = (int)list[have ? idx : 0];
        const int oly = ptid >> 4, olx = 2 * (ptid & 15) + ((oly + colour) & 1);
        const int npx = x0 + olx, npy = y0 + oly;
        const float4 pl = norm4[npy * cols + npx];
        const int tnx = olx + halo, tny = oly + halo;
        const float nxf = (float)npx, nyf = (float)npy;
        // compressed-tile indices: this lane's first point of either family, its consumer's centre
        // and window corner
        const int base_v = tny * twc + ((tnx - R) >> 1) + l;
        const int base_h = tny * twc + ((tnx - (R + kPushReach)) >> 1) + l;
        const int cpx = npx + cdx, cpy = npy + cdy;
        const bool cvalid = have && cpx >= 0 && cpx < cols && cpy >= 0 && cpy < rows;
        const int tpx = tnx + cdx, tpy = tny + cdy;
        const float centre = iplane[tpy * twc + (tpx >> 1)];
        const float *ipl = iplane + (tpy - R) * twc + ((tpx - R) >> 1);

        ViewCombiner<true> comb;
        // the homography of a (plane, view) pair is the same for the lanes of a group: lane c computes
        // it for view vb + c (the literal arithmetic of homography()), the lanes then pass them round
        for (int vb = 0; vb < n; vb += kPushLanes) {
            float Hl[9];
            homography(P->rc.K_inv, P->view[min(vb + l, n - 1)], pl, Hl);
            const int vend = min(vb + kPushLanes, n);
            for (int v = vb; v < vend; v++) {
                float H[9];
#pragma unroll
                for (int k = 0; k < 9; k++) H[k] = __shfl(Hl[k], grp_lane0 + (v - vb));
                // 1/Z by rcp + Newton where the whole stencil is provably inside its exact range
                // (any such proof gives the bits of the IEEE division, see rcp_newton)
                const float reach = (float)(R + kPushReach);
                const bool safe = window_z_safe(H, nxf - reach, nxf + reach, nyf - reach, nyf + reach);
                const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)P->view[v].packed - (uintptr_t)kMagicBits);
                if (__all(safe))
                    push_eval<BOX, true>(P, magic_base, H, tile4, base_v, base_h, dgrp, nxf, nyf, l);
                else
                    push_eval<BOX, false>(P, magic_base, H, tile4, base_v, base_h, dgrp, nxf, nyf, l);
                __builtin_amdgcn_wave_barrier();  // (the group's samples are read by other lanes of this wavefront)
                const float c = push_chain<BOX>(ipl, centre, lut_magic, dgrp + dbase, jstride);
                __builtin_amdgcn_wave_barrier();
                comb.add(c, v, nullptr);
            }
        }
        const float F = comb.finish(P, n, nullptr);
        if (cvalid) P->push_cost[(size_t)l * np + (size_t)(cpy * cols + cpx)] = F;
    }
}

}  // namespace pm
