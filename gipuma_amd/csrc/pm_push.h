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
// Supported: gray window-packed planes with float-encoded offsets, box 11 / 15 / 25, best-N with
// n_best <= 4.  Box 15 (8 window columns = 8 lanes) has its steps unrolled with compile-time stencil
// positions; the other boxes run PushEval::family.  Colour (T = float4, box 15): push_kernel_c4 below.
// What binds it (config C, profiles/r02_push_pmc_push_kernel.json): the vector L1 -- 1.39e9 accesses per
// launch, and its miss handling while the planes are random -- not VALU issue: a timing build without
// the chain phase is no faster.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int kPushReach = 5;                         // propagation distance (gipuma.cu:1437-1462)
constexpr int kPushLanes = 8;                         // lanes per producer
constexpr int kPushGroups = kThreads / kPushLanes;    // producers evaluated concurrently by a workgroup
// The producers of a tile are dealt to the lane groups ordered by DISPARITY bucket (performance only: a producer's eight
// costs do not depend on when it is evaluated).  While the planes are random, the 32 producers a workgroup evaluates
// together then see the source views at similar offsets along the epipolar lines: their windows fall into a region the
// CU's vector L1 holds (a sixteenth of the disparity range plus the tile: ~20 KB per view) instead of across the whole
// range (~700 px on config C: 120 KB).
#ifndef PM_DISP_BUCKETS
#define PM_DISP_BUCKETS 16
#endif
constexpr int kPushBuckets = PM_DISP_BUCKETS;
__device__ __forceinline__ int disparity_bucket(const Problem *__restrict__ P, float4 pl, int px, int py)
{
    const float depth = depth_from_plane(P->rc, pl, px, py);
    const float disp = disp_depth(P->rc.f, P->rc.baseline, depth);
    const float t = (disp - P->min_disp) / (P->max_disp - P->min_disp) * (float)kPushBuckets;
    return (int)__builtin_fminf(__builtin_fmaxf(t, 0.0f), (float)(kPushBuckets - 1));  // (NaN -> 0)
}

template <int BOX>
struct PushLayout {  // offsets in 32-bit words into the dynamic LDS array
    static_assert(BOX == 11 || BOX == 15 || BOX == 19 || BOX == 25, "instantiated window sizes");
    static constexpr int R = (BOX - 1) / 2, N = R + 1;
    static constexpr int FWH = N + kPushReach;        // box 15: 13 rows of the vertical family / columns of the horizontal one
    static constexpr int NF = N * FWH;                // 104 points per family
    static constexpr int halo = R + kPushReach + 1;   // 13: samples reach 12 texels, their gradients one more
    static constexpr int tw = kTileW + 2 * halo, th = kSweepTileH + 2 * halo;  // 58 x 42
    // only texels of the consumers' colour are ever sampled (producer + odd + even offsets): the tile
    // is stored checkerboard-compressed, entry (ty, tx >> 1)
    static constexpr int twc = ((tw + 1) / 2) | 1;    // 29 (odd: rows land in different banks)
    // box 25: 2 x 234 samples per group would leave one workgroup per CU -- the group's buffer holds
    // one family at a time, and the chain phase runs once per family (half of its lanes idle)
    static constexpr bool two_pass = BOX > 15;
    // The group's sample buffer: rows of the vertical family `vs` words apart, of the horizontal one `hs`,
    // the horizontal family at `hbase`, groups `dstride` apart; the I plane's rows `ips` apart.  Box 15
    // (unrolled loop): strides searched with the bank model so that EVERY read of the chain phase -- 32
    // lanes = 4 producers x 8 consumers per LDS cycle -- is conflict-free (scripts/exp/push_banks.py;
    // the dense layout 8 / 13 / 104 / 210 / 29 is 2-way everywhere: 47 % of the kernel's LDS cycles).
#ifdef PM_PUSH_GENERIC15
    static constexpr bool tuned = false;
#else
    static constexpr bool tuned = BOX == 15;
#endif
    static constexpr int vs = tuned ? 9 : N, hs = tuned ? 17 : FWH;
    static constexpr int hbase = two_pass ? 0 : tuned ? 124 : NF;   // horizontal family behind the vertical one
    static constexpr int dstride = tuned ? 264 : BOX == 15 ? 2 * NF + 2 : BOX == 11 ? 2 * NF + 4 : NF + 2;
    static constexpr int ips = tuned ? 41 : twc;
    static_assert(FWH * vs <= (two_pass ? dstride : hbase) && hbase + N * hs <= dstride, "families inside the buffer");
    static constexpr int tile4 = kLutSize;            // {I, gx1, gy1, I} per compressed texel
    static constexpr int iplane = tile4 + 4 * twc * th;  // I alone (4-byte reads of the chain)
#ifdef PM_PUSH_EXP_ALIAS  // (timing experiment, WRONG results: the sample buffers on top of the tile, so that three workgroups fit a CU)
    static constexpr int dis = tile4;
#else
    static constexpr int dis = iplane + ips * th;     // [kPushGroups][dstride]; before that the staging plane
#endif
    static constexpr int list = dis + kPushGroups * dstride;  // 256 u16: producers with something to offer
    static constexpr int cnt = list + kThreads / 2;   // [4] live producers per wavefront, [8 + kPushBuckets] bucket counters
    static constexpr int total = cnt + 8 + 64;
    static_assert(tw * th <= kPushGroups * dstride, "the staging plane aliases the sample buffers");
    static_assert(total * 4 <= 80 * 1024, "two workgroups per CU");
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
// (scripts/exp/push_model.py checks these index formulas against the definition.)  kPushPD window
// requests are in flight per lane; the first kPushPD of a view are issued by first() BEFORE the
// chain phase of the previous view, so that their latency -- L2 misses while the planes are random --
// hides behind it.
// Measured on config C (first three half-sweeps, same box; scripts/exp/ab_push*.txt): what matters is
// that the loop neither spills nor starves the scheduler of registers -- 3 wavefronts per SIMD
// (168 VGPRs, 70-130 scratch accesses per view competing with the window loads for the vector L1)
// 16.2 / 9.0 / 8.4 ms, 2 wavefronts per SIMD (234 VGPRs, no scratch) 13.4 / 8.0 / 7.7 ms; prefetch depth
// 2..8 and cross-view prefetch make no difference there, 10-12 lose, one workgroup per CU loses 35 %.
#ifndef PM_PUSH_PD
#define PM_PUSH_PD 4
#endif
#ifndef PM_PUSH_WAVES
#define PM_PUSH_WAVES 2
#endif
constexpr int kPushPD = PM_PUSH_PD;
#ifndef PM_PUSH_PD_FAMILY
#define PM_PUSH_PD_FAMILY 2
#endif
constexpr int kPushPDFamily = PM_PUSH_PD_FAMILY;  // PushEval::family (boxes 11, 25; colour): 2 / 4 / 8 within 1.5 % on config D, 2 best
template <int BOX>
struct PushEval {
    using LY = PushLayout<BOX>;
    static constexpr int FWH = LY::FWH, twc = LY::twc, S = 2 * FWH, PD = kPushPD;
    MagicAddr MA;       // wave-uniform
    DisConst K;         // (alpha / 16 and 16 tau_g: dis_fold, pm_sample.h)
    const Problem *P;
    float qx_v, qx_h, nyf;  // (float)(n + d) == (float)n + (float)d exactly (small integers)
    int l, base_v, base_h;

    __device__ __forceinline__ void init(const Problem *__restrict__ P_, int lane_in_group)
    {
        P = P_;
        MA = magic_addr(P_);
        K = dis_const(P_);
        l = lane_in_group;
    }
    __device__ __forceinline__ void producer(float nxf, float nyf_, int tnx, int tny)
    {
        qx_v = nxf + (float)(2 * l - LY::R);                 // vertical family: the lane's column
        qx_h = nxf + (float)(2 * l - (LY::R + kPushReach));  // horizontal family, before the row wrap
        nyf = nyf_;
        // compressed-tile indices of this lane's first point of either family
        base_v = tny * twc + ((tnx - LY::R) >> 1) + l;
        base_h = tny * twc + ((tnx - (LY::R + kPushReach)) >> 1) + l;
    }
    template <bool FAST>
    __device__ __forceinline__ WinReq request(const float *__restrict__ H, gptr_bytes magic_base, float qx, float qy) const
    {
        // getCorrespondingPoint_cu, gipuma.cu:207-217, the arithmetic of view_cost_pipe (pm_sample.h)
#ifdef PM_PUSH_EXP_ONE_WINDOW_PER_GROUP  // (timing experiment, wrong results: the 8 lanes of a group fetch ONE window --
        qx = qx - (float)(2 * l);        //  the vector L1's cost if a group's window bytes were fetched once)
#endif
        return magic_request<FAST>(MA, magic_base, warp_col(H, qx), warp_row(H), qy);
    }
    template <bool FAST>
    __device__ __forceinline__ WinReq issue(const float *__restrict__ H, gptr_bytes magic_base, int s) const
    {
        if (s < FWH) return request<FAST>(H, magic_base, qx_v, nyf + (float)(2 * s - (LY::R + kPushReach)));
        const int t = s - FWH;
        const int j0 = (8 * t) / FWH, r0 = (8 * t) % FWH;
        const bool wrap = (FWH - r0 < kPushLanes) && l >= FWH - r0;
        const float qx = qx_h + (wrap ? (float)(2 * r0 - 2 * FWH) : (float)(2 * r0));
        const float qy = nyf + (wrap ? (float)(2 * j0 + 2 - LY::R) : (float)(2 * j0 - LY::R));
        return request<FAST>(H, magic_base, qx, qy);
    }
    __device__ __forceinline__ int tile_index(int s) const
    {
        if (s < FWH) return base_v + (2 * s - (LY::R + kPushReach)) * twc;
        const int t = s - FWH;
        const int j0 = (8 * t) / FWH, r0 = (8 * t) % FWH;
        const bool wrap = (FWH - r0 < kPushLanes) && l >= FWH - r0;
        return base_h + (2 * j0 - LY::R) * twc + r0 + (wrap ? 2 * twc - FWH : 0);
    }
    // where the sample of step s goes in the group's buffer (row strides LY::vs / LY::hs)
    __device__ __forceinline__ int slot(int s) const
    {
        if (s < FWH) return s * LY::vs + l;
        const int t = s - FWH;
        const int j0 = (8 * t) / FWH, r0 = (8 * t) % FWH;
        const bool wrap = (FWH - r0 < kPushLanes) && l >= FWH - r0;
        return LY::hbase + j0 * LY::hs + r0 + l + (wrap ? LY::hs - FWH : 0);
    }
    // the first PD requests of a view
    template <bool FAST>
    __device__ __forceinline__ void first(const float *__restrict__ H, gptr_bytes magic_base, WinReq (&req)[kPushPD]) const
    {
#pragma unroll
        for (int p = 0; p < PD; p++) req[p] = issue<FAST>(H, magic_base, p);
    }
    // all S steps of a view whose first PD requests are in `req`
    template <bool FAST>
    __device__ __forceinline__ void body(const float *__restrict__ H, gptr_bytes magic_base, WinReq (&req)[kPushPD],
                                         const float *__restrict__ tile4, float *__restrict__ dgrp) const
    {
#pragma unroll
        for (int s = 0; s < S; s++) {
            const WinReq cur = req[s % PD];
            if (s + PD < S) req[s % PD] = issue<FAST>(H, magic_base, s + PD);
            // {I(q), gx1(q), gy1(q), I(q)} of the reference tile
            const float4 t4 = *reinterpret_cast<const float4 *>(tile4 + 4 * tile_index(s));
            // the taps and pmCostComputation_shared, gipuma.cu:251-274
            dgrp[slot(s)] = gray_dis(K, cur, t4.w, t4.y, t4.z, plane_of_magic(magic_base, P));
#ifndef PM_PUSH_NO_SCHED_BARRIER
            // keep the steps apart: left alone, the scheduler interleaves many of them and spills
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }

    // ---- any box: one family of the stencil, W points per row, HH rows, first point at offset
    //      (-OX, -OY) from the producer, dealt 8 consecutive points (row-major) per step.  The stream
    //      position (row j0, first column r0 of a step) is wave-uniform; a lane is one or -- rows
    //      shorter than 8 points -- two rows further when r0 + l runs past the row end.  Steps past the
    //      end of the family (the prefetch) and the spare lanes of its last step compute harmless
    //      points just outside it and store nothing. ----
    template <bool FAST, int W, int HH, int OX, int OY>
    __device__ __forceinline__ void family(const float *__restrict__ H, gptr_bytes magic_base,
                                           const float *__restrict__ tile4, float *__restrict__ dfam, float nxf,
                                           int tnx, int tny) const
    {
        constexpr int NFp = W * HH, SF = (NFp + kPushLanes - 1) / kPushLanes;
        static_assert(2 * W >= kPushLanes, "at most two row wraps per step");
        const float qx0 = nxf + (float)(2 * l - OX);
        const float qy0 = nyf - (float)OY;
        const int tbase = (tny - OY) * twc + ((tnx - OX) >> 1) + l;
        struct Pos {
            int r0, j0;
        };
        auto advance = [](Pos &p) {
            p.r0 += kPushLanes;
            if (p.r0 >= W) {
                p.r0 -= W;
                p.j0++;
            }
            if (W < kPushLanes && p.r0 >= W) {
                p.r0 -= W;
                p.j0++;
            }
        };
        auto wraps = [&](const Pos &p, bool &w1, bool &w2) {
            w1 = l >= W - p.r0;
            w2 = W < kPushLanes && l >= 2 * W - p.r0;
        };
        auto issue_at = [&](const Pos &p) -> WinReq {
            bool w1, w2;
            wraps(p, w1, w2);
            const float qx = qx0 + (float)(2 * p.r0) - (w1 ? (float)(2 * W) : 0.0f) - (w2 ? (float)(2 * W) : 0.0f);
            const float qy = qy0 + (float)(2 * p.j0) + (w1 ? 2.0f : 0.0f) + (w2 ? 2.0f : 0.0f);
            return request<FAST>(H, magic_base, qx, qy);
        };
        auto tile_at = [&](const Pos &p) -> int {
            bool w1, w2;
            wraps(p, w1, w2);
            return tbase + p.r0 + 2 * twc * p.j0 + (w1 ? 2 * twc - W : 0) + (w2 ? 2 * twc - W : 0);
        };
        constexpr int PD = kPushPDFamily;
        Pos pi{0, 0}, pr{0, 0};
        WinReq req[PD];
#pragma unroll
        for (int p = 0; p < PD; p++) {
            req[p] = issue_at(pi);
            advance(pi);
        }
        // (fully unrolled since round 6: the stream positions, the wrap tests and the tile offsets become compile-time
        //  constants -- config D 465.9 -> 448.3 ms per view, box 19 134.4 -> 133.7; -DPM_PUSH_FAMILY_ROLLED: the rolled loop)
#ifdef PM_PUSH_FAMILY_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int sb = 0; sb < SF; sb += PD) {
#pragma unroll
            for (int p = 0; p < PD; p++) {
                const int s = sb + p;
                const WinReq cur = req[p];
                req[p] = issue_at(pi);  // (unconditional: clamped, valid addresses past the end, dropped)
                advance(pi);
                if (s < SF) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(tile4 + 4 * tile_at(pr));
                    const float dv = gray_dis(K, cur, t4.w, t4.y, t4.z, plane_of_magic(magic_base, P));
                    const int e = kPushLanes * s + l;
                    if (e < NFp) dfam[e] = dv;
                }
                advance(pr);
#ifndef PM_PUSH_NO_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    }
    template <bool FAST>
    __device__ __forceinline__ void family_v(const float *__restrict__ H, gptr_bytes mb, const float *__restrict__ tile4,
                                             float *__restrict__ dfam, float nxf, int tnx, int tny) const
    {
        family<FAST, LY::N, FWH, LY::R, LY::R + kPushReach>(H, mb, tile4, dfam, nxf, tnx, tny);
    }
    template <bool FAST>
    __device__ __forceinline__ void family_h(const float *__restrict__ H, gptr_bytes mb, const float *__restrict__ tile4,
                                             float *__restrict__ dfam, float nxf, int tnx, int tny) const
    {
        family<FAST, FWH, LY::N, LY::R + kPushReach, LY::R>(H, mb, tile4, dfam, nxf, tnx, tny);
    }
};

// the reference's summation for one consumer: columns outer, rows inner, one accumulation per sample
// (gipuma.cu:633-676); `ipl` = the compressed I plane at the consumer's window corner, `dch` = the
// group's sample buffer at the consumer's first sample, rows `jstride` apart
template <int BOX>
__device__ __forceinline__ float push_chain(const float *__restrict__ ipl, float centre, const char *lut_magic,
                                            const float *__restrict__ dch, int jstride)
{
    using LY = PushLayout<BOX>;
    float cost = 0.0f;
#ifdef PM_PUSH_CHAIN_UNROLLED
#pragma unroll
#else
#pragma unroll 1  // (8 terms per iteration; all 64 unrolled: +12 KB of code, no faster)
#endif
    for (int i = 0; i < LY::N; i++)
#pragma unroll
        for (int j = 0; j < LY::N; j++) {
            // weight_cu, gipuma.cu:186-193: 256 possible weights
            const float w = lut_weight(lut_magic, ipl[2 * j * LY::ips + i], centre);
            cost = accum(w, dch[j * jstride + i], cost);
        }
    return cost;
}

// grid = the sweep tiles of the frame; `colour` = the colour of the producers (the colour that was
// swept last); hist: offer only the planes that changed in that half-sweep (rule (H))
template <int BOX>
__global__ __launch_bounds__(kThreads, PM_PUSH_WAVES) void push_kernel(const Problem *__restrict__ P,
                                                           const float4 *__restrict__ norm4, int colour, int hist,
                                                           unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = PushLayout<BOX>;
    constexpr int R = LY::R, twc = LY::twc, halo = LY::halo;
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
    const bool live = inside && (!hist || P->changed[PM_AT(P, py * cols + px, PM_NP(P), kChkFlags)] != 0);
    const unsigned long long bal = __ballot(live);
    if (lane == 0) cnt[wave] = (int)__popcll(bal);
    const bool by_disp = !(tune & Tune::kNoDispSort);
    if (tid < kPushBuckets) cnt[8 + tid] = 0;
    const int bucket = live && by_disp ? disparity_bucket(P, norm4[PM_AT(P, py * cols + px, PM_NP(P), kChkNorm4)], px, py) : 0;

    // ---- reference tile (clamp-to-edge point samples like the reference's, gipuma.cu:1393-1402),
    //      checkerboard-compressed, with the gradients of pmCostComputation_shared (:254-259) ----
    {
        constexpr int tw = LY::tw, th = LY::th;
        const gptr_f32 ref = (gptr_f32)P->ref.raw;
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
            iplane[ty * LY::ips + cx] = I;
        }
    }
    const int c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];  // (written before the first barrier above)
    const int n_live = c0 + c1 + c2 + c3;
    if (by_disp) {
        // counting sort by bucket: rank inside the bucket from an LDS counter, bucket starts from a 16-entry scan
        int rank_b = 0;
        if (live) rank_b = atomicAdd(&cnt[8 + bucket], 1);
        __syncthreads();
        if (live) {
            int start = 0;
#pragma unroll
            for (int b = 0; b < kPushBuckets; b++) start += b < bucket ? cnt[8 + b] : 0;
            list[start + rank_b] = (unsigned short)tid;
        }
    } else if (live) {
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
    const int dbase = cdx == 0 ? ((cdy + kPushReach) / 2) * LY::vs : LY::hbase + (cdx + kPushReach) / 2;
    const int jstride = cdx == 0 ? LY::vs : LY::hs;
    PushEval<BOX> E;
    E.init(P, l);

    for (int r = 0; r * kPushGroups < n_live; r++) {
        const int idx = r * kPushGroups + grp;
        const bool have = idx < n_live;
        if (!__any(have)) break;  // (later wavefronts hold the larger indices)
        const int ptid = (int)list[have ? idx : 0];
        const int oly = ptid >> 4, olx = 2 * (ptid & 15) + ((oly + colour) & 1);
        const int npx = x0 + olx, npy = y0 + oly;
        const float4 pl = norm4[PM_AT(P, npy * cols + npx, np, kChkNorm4)];
        const int tnx = olx + halo, tny = oly + halo;
        const float nxf = (float)npx, nyf = (float)npy;
        E.producer(nxf, nyf, tnx, tny);
        // this lane's consumer: its centre and window corner in the compressed I plane
        const int cpx = npx + cdx, cpy = npy + cdy;
        const bool cvalid = have && cpx >= 0 && cpx < cols && cpy >= 0 && cpy < rows;
        const int tpx = tnx + cdx, tpy = tny + cdy;
        const float centre = iplane[tpy * LY::ips + (tpx >> 1)];
        const float *ipl = iplane + (tpy - R) * LY::ips + ((tpx - R) >> 1);

        ViewCombiner<true> comb;
        // The homography of a (plane, view) pair is the same for the lanes of a group: lane c computes
        // it for view 8b + c (the literal arithmetic of homography()), the lanes then pass them round.
        // 1/Z by rcp + Newton where the whole stencil is provably inside its exact range (any such
        // proof gives the bits of the IEEE division, see rcp_newton).
        float Hl[9], H[9];
        bool fast = false;
        gptr_bytes magic_base = nullptr;
        auto next_h = [&](int v) {
#ifdef PM_PUSH_DIRECT_H
            homography(P->rc.K_inv, P->view[v], pl, H);  // (experiment: every lane, no exchange)
            (void)Hl;
            (void)grp_lane0;
#else
            if ((v & (kPushLanes - 1)) == 0) homography(P->rc.K_inv, P->view[min(v + l, n - 1)], pl, Hl);
#pragma unroll
            for (int k = 0; k < 9; k++) H[k] = __shfl(Hl[k], grp_lane0 + (v & (kPushLanes - 1)));
#endif
            const float reach = (float)(R + kPushReach);
            fast = __all(window_z_safe(H, nxf - reach, nxf + reach, nyf - reach, nyf + reach));
            magic_base = (gptr_bytes)((uintptr_t)P->view[v].packed.raw - (uintptr_t)kMagicBits);
        };
#ifdef PM_PUSH_GENERIC15
        if constexpr (false) {
#else
        if constexpr (BOX == 15) {
#endif
            WinReq req[kPushPD];
            auto load_view = [&](int v) {
                next_h(v);
                if (fast)
                    E.template first<true>(H, magic_base, req);
                else
                    E.template first<false>(H, magic_base, req);
            };
            load_view(0);
            for (int v = 0; v < n; v++) {
                if (fast)
                    E.template body<true>(H, magic_base, req, tile4, dgrp);
                else
                    E.template body<false>(H, magic_base, req, tile4, dgrp);
                if (v + 1 < n) load_view(v + 1);  // the next view's first windows travel during the chain
                __builtin_amdgcn_wave_barrier();  // (the group's samples are read by other lanes of this wavefront)
#ifdef PM_PUSH_EXP_NOCHAIN  // timing experiment only (wrong results): what the chain phase costs
                const float c = dgrp[dbase + (v & 63)];
#else
                const float c = push_chain<BOX>(ipl, centre, lut_magic, dgrp + dbase, jstride);
#endif
                __builtin_amdgcn_wave_barrier();
                comb.add(c, v, nullptr);
            }
        } else {
            for (int v = 0; v < n; v++) {
                next_h(v);
                float c;
                if constexpr (LY::two_pass) {
                    if (fast)
                        E.template family_v<true>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                    else
                        E.template family_v<false>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                    __builtin_amdgcn_wave_barrier();
                    // (only the lanes whose consumer belongs to the family in the buffer: the others would
                    //  read values they do not use, and their reads collide with the useful ones in the
                    //  LDS banks -- measured: the box-25 kernel was LDS-bound, half of it bank conflicts)
                    float cv_ = 0.0f, ch_ = 0.0f;
                    if (cdx == 0) cv_ = push_chain<BOX>(ipl, centre, lut_magic, dgrp + dbase, jstride);
                    __builtin_amdgcn_wave_barrier();
                    if (fast)
                        E.template family_h<true>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                    else
                        E.template family_h<false>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                    __builtin_amdgcn_wave_barrier();
                    if (cdx != 0) ch_ = push_chain<BOX>(ipl, centre, lut_magic, dgrp + dbase, jstride);
                    __builtin_amdgcn_wave_barrier();
                    c = cdx == 0 ? cv_ : ch_;
                } else {
                    if (fast) {
                        E.template family_v<true>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                        E.template family_h<true>(H, magic_base, tile4, dgrp + LY::hbase, nxf, tnx, tny);
                    } else {
                        E.template family_v<false>(H, magic_base, tile4, dgrp, nxf, tnx, tny);
                        E.template family_h<false>(H, magic_base, tile4, dgrp + LY::hbase, nxf, tnx, tny);
                    }
                    __builtin_amdgcn_wave_barrier();
                    c = push_chain<BOX>(ipl, centre, lut_magic, dgrp + dbase, jstride);
                    __builtin_amdgcn_wave_barrier();
                }
                comb.add(c, v, nullptr);
            }
        }
        const float F = comb.finish(P, n, nullptr);
        if (cvalid) P->push_cost[PM_AT(P, (size_t)l * np + (size_t)(cpy * cols + cpx), 8 * np, kChkPushCost)] = F;
    }
}

// ---------------------------------------------------------------------------------------------
// The same kernel for -color_processing (T = float4, gipuma.cu:1965-1968; view_cost_c4_loop in
// pm_device.h): three 16-byte window loads and three tap sets per stencil point, l1_norm(float4)
// reductions in the reference's order, support weights from the 766-entry table indexed by the
// integer |dB|+|dG|+|dR|.  Integer window addressing (three words per texel do not fit the
// float-encoded offsets).  The reference tile holds, per texel of the consumers' colour, {B, G, R}
// and the central differences right-left, down-up of each channel (the reference-side terms of
// pmCostComputation_shared, gipuma.cu:254-259) -- the neighbours themselves have the other parity and
// are not kept.  Box 15, packed 8-bit planes, best-N with n_best <= 4.
// ---------------------------------------------------------------------------------------------
template <int BOX>
struct PushLayoutC4 {
    static_assert(BOX == 15, "instantiated window size");
    using G = PushLayout<BOX>;
    static constexpr int R = G::R, N = G::N, FWH = G::FWH, NF = G::NF, halo = G::halo, tw = G::tw, th = G::th, twc = G::twc;
    static constexpr int hbase = NF, dstride = 2 * NF + 2;  // dense rows (N / FWH words), the generic stencil loop
    static constexpr int tile_a = lut_size<4>();           // float4 {B, G, R, gxB} per compressed texel
    static constexpr int tile_b = tile_a + 4 * twc * th;   // float4 {gxG, gxR, gyB, gyG}
    static constexpr int tile_c = tile_b + 4 * twc * th;   // float gyR
    static constexpr int dis = tile_c + twc * th;
    static constexpr int list = dis + kPushGroups * dstride;
    static constexpr int cnt = list + kThreads / 2;   // [4] live producers per wavefront, [8 + kPushBuckets] bucket counters
    static constexpr int total = cnt + 8 + 64;
    static_assert(total * 4 <= 80 * 1024, "two workgroups per CU");
};

template <int BOX>
struct PushEvalC4 {
    using LY = PushLayoutC4<BOX>;
    static constexpr int FWH = LY::FWH, twc = LY::twc;
    DisConst K;  // (alpha / 16 and 16 tau_g: dis_fold, pm_sample.h)
    IntAddr IA;
    const Problem *P;
    float nyf;
    int l;

    __device__ __forceinline__ void init(const Problem *__restrict__ P_, int lane_in_group)
    {
        P = P_;
        K = dis_const(P_);
        IA = int_addr(P_);
        l = lane_in_group;
    }
    template <bool FAST>
    __device__ __forceinline__ WinReq3 request(const float *__restrict__ H, gptr_bytes packed, float qx, float qy) const
    {
        return c4_request<FAST>(IA, packed, warp_col(H, qx), warp_row(H), qy);
    }
    // one family of the stencil: the stream of PushEval::family with the colour request / reduction
    template <bool FAST, int W, int HH, int OX, int OY>
    __device__ __forceinline__ void family(const float *__restrict__ H, gptr_bytes packed, const float *__restrict__ lds,
                                           float *__restrict__ dfam, float nxf, int tnx, int tny) const
    {
        constexpr int NFp = W * HH, SF = (NFp + kPushLanes - 1) / kPushLanes;
        static_assert(W >= kPushLanes, "one row wrap per step");
        const float qx0 = nxf + (float)(2 * l - OX);
        const float qy0 = nyf - (float)OY;
        const int tbase = (tny - OY) * twc + ((tnx - OX) >> 1) + l;
        struct Pos {
            int r0, j0;
        };
        auto advance = [](Pos &p) {
            p.r0 += kPushLanes;
            if (p.r0 >= W) {
                p.r0 -= W;
                p.j0++;
            }
        };
        auto issue_at = [&](const Pos &p) -> WinReq3 {
            const bool w1 = l >= W - p.r0;
            const float qx = qx0 + (float)(2 * p.r0) - (w1 ? (float)(2 * W) : 0.0f);
            const float qy = qy0 + (float)(2 * p.j0) + (w1 ? 2.0f : 0.0f);
            return request<FAST>(H, packed, qx, qy);
        };
        auto tile_at = [&](const Pos &p) -> int {
            const bool w1 = l >= W - p.r0;
            return tbase + p.r0 + 2 * twc * p.j0 + (w1 ? 2 * twc - W : 0);
        };
        constexpr int PD = kPushPDFamily;
        Pos pi{0, 0}, pr{0, 0};
        WinReq3 req[PD];
#pragma unroll
        for (int p = 0; p < PD; p++) {
            req[p] = issue_at(pi);
            advance(pi);
        }
#ifdef PM_PUSH_FAMILY_ROLLED  // (unrolled since round 6 like PushEval::family: colour 182.2 -> 177.7 ms per view)
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int sb = 0; sb < SF; sb += PD) {
#pragma unroll
            for (int p = 0; p < PD; p++) {
                const int s = sb + p;
                const WinReq3 cur = req[p];
                req[p] = issue_at(pi);  // (unconditional: clamped, valid addresses past the end, dropped)
                advance(pi);
                if (s < SF) {
                    const int k = tile_at(pr);
                    const float4 ta = *reinterpret_cast<const float4 *>(lds + LY::tile_a + 4 * k);  // B, G, R, gxB
                    const float4 tb = *reinterpret_cast<const float4 *>(lds + LY::tile_b + 4 * k);  // gxG, gxR, gyB, gyG
                    const float tc = lds[LY::tile_c + k];                                           // gyR
                    // word 3k+c = column k, channel c (view_cost_c4_loop)
                    Taps t[3];
                    c4_taps(cur, plane_of(packed, P), t);
                    // pmCostComputation_shared for T = float4, gipuma.cu:251-274
                    const float colDiff = l1_3(ta.x - t[0].sc, ta.y - t[1].sc, ta.z - t[2].sc);
                    const float gX = l1_3(ta.w - t[0].gx2, tb.x - t[1].gx2, tb.y - t[2].gx2);
                    const float gY = l1_3(tb.z - t[0].gy2, tb.w - t[1].gy2, tc - t[2].gy2);
                    const int e = kPushLanes * s + l;
                    if (e < NFp) dfam[e] = dis_folded<false>(gX + gY, colDiff, K.alpha16, K.oma, K.tau_color, K.taug16);
                }
                advance(pr);
#ifndef PM_PUSH_NO_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    }
};

// the reference's summation for one consumer, T = float4: weight table indexed by |dB|+|dG|+|dR|
template <int BOX>
__device__ __forceinline__ float push_chain_c4(const float *__restrict__ ta, float4 centre, const float *__restrict__ lut,
                                               const float *__restrict__ dch, int jstride)
{
    using LY = PushLayoutC4<BOX>;
    float cost = 0.0f;
#pragma unroll 1
    for (int i = 0; i < LY::N; i++)
#pragma unroll
        for (int j = 0; j < LY::N; j++) {
            const float4 lv = *reinterpret_cast<const float4 *>(ta + 4 * (2 * j * LY::twc + i));
            const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                            __builtin_fabsf(lv.z - centre.z);  // exact integer 0..765
            cost = accum(lut[(int)S], dch[j * jstride + i], cost);
        }
    return cost;
}

template <int BOX>
__global__ __launch_bounds__(kThreads, PM_PUSH_WAVES) void push_kernel_c4(const Problem *__restrict__ P,
                                                              const float4 *__restrict__ norm4, int colour, int hist,
                                                              unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = PushLayoutC4<BOX>;
    constexpr int R = LY::R, N = LY::N, twc = LY::twc, halo = LY::halo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = P->rows, cols = P->cols;
    const int gx = (cols + kTileW - 1) / kTileW;
    const int gy = (rows + kSweepTileH - 1) / kSweepTileH;
    const TileXY txy = tile_of(blockIdx.x, gx, gy, tune);
    const int x0 = txy.x * kTileW, y0 = txy.y * kSweepTileH;
    unsigned short *list = reinterpret_cast<unsigned short *>(lds + LY::list);
    int *cnt = reinterpret_cast<int *>(lds + LY::cnt);

    // ---- producers of this tile with something to offer ----
    const int ly = tid >> 4, lx = 2 * (tid & 15) + ((ly + colour) & 1);
    const int px = x0 + lx, py = y0 + ly;
    const bool inside = px < cols && py < rows;
    const bool live = inside && (!hist || P->changed[PM_AT(P, py * cols + px, PM_NP(P), kChkFlags)] != 0);
    const unsigned long long bal = __ballot(live);
    if (lane == 0) cnt[wave] = (int)__popcll(bal);
    const bool by_disp = !(tune & Tune::kNoDispSort);
    if (tid < kPushBuckets) cnt[8 + tid] = 0;
    const int bucket = live && by_disp ? disparity_bucket(P, norm4[PM_AT(P, py * cols + px, PM_NP(P), kChkNorm4)], px, py) : 0;

    // ---- reference tile: texels of the consumers' colour, clamp-to-edge point samples like the
    //      reference's (gipuma.cu:1393-1402), each with its channel-wise central differences ----
    {
        const gptr_f32 ref = (gptr_f32)P->ref.raw;
        const int pitch = P->pitch;
        const int cpar = 1 - colour;
        for (int k = tid; k < lut_size<4>(); k += kThreads) lds[k] = exp_model(-((float)k * 0.3333333f) / P->gamma);
        for (int k = tid; k < twc * LY::th; k += kThreads) {
            const int ty = k / twc, cx = k - ty * twc;
            const int tx = 2 * cx + ((cpar + ty) & 1);
            const int gxp = x0 - halo + tx, gyp = y0 - halo + ty;
            const int xc = clampi(gxp, 0, cols - 1), xl = clampi(gxp - 1, 0, cols - 1), xr = clampi(gxp + 1, 0, cols - 1);
            const int yc = clampi(gyp, 0, rows - 1), yu = clampi(gyp - 1, 0, rows - 1), yd = clampi(gyp + 1, 0, rows - 1);
            const gptr_f32 c = ref + (yc * pitch + 4 * xc);
            const gptr_f32 le = ref + (yc * pitch + 4 * xl), ri = ref + (yc * pitch + 4 * xr);
            const gptr_f32 up = ref + (yu * pitch + 4 * xc), dn = ref + (yd * pitch + 4 * xc);
            *reinterpret_cast<float4 *>(lds + LY::tile_a + 4 * k) = make_float4(c[0], c[1], c[2], ri[0] - le[0]);
            *reinterpret_cast<float4 *>(lds + LY::tile_b + 4 * k) =
                make_float4(ri[1] - le[1], ri[2] - le[2], dn[0] - up[0], dn[1] - up[1]);
            lds[LY::tile_c + k] = dn[2] - up[2];
        }
    }
    __syncthreads();
    const int c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];
    const int n_live = c0 + c1 + c2 + c3;
    if (by_disp) {  // (producers ordered by disparity bucket, as in push_kernel)
        int rank_b = 0;
        if (live) rank_b = atomicAdd(&cnt[8 + bucket], 1);
        __syncthreads();
        if (live) {
            int start = 0;
#pragma unroll
            for (int b = 0; b < kPushBuckets; b++) start += b < bucket ? cnt[8 + b] : 0;
            list[start + rank_b] = (unsigned short)tid;
        }
    } else if (live) {
        const int first = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        list[first + rank] = (unsigned short)tid;
    }
    __syncthreads();
    if (n_live == 0) return;

    const int n = P->n_sel;
    const size_t np = (size_t)rows * (size_t)cols;
    const int grp = tid / kPushLanes, l = tid % kPushLanes;
    const int grp_lane0 = lane & ~(kPushLanes - 1);
    float *dgrp = lds + LY::dis + grp * LY::dstride;
    int cdx, cdy;
    push_consumer_offset(l, cdx, cdy);
    const int dbase = cdx == 0 ? ((cdy + kPushReach) / 2) * N : LY::hbase + (cdx + kPushReach) / 2;
    const int jstride = cdx == 0 ? N : LY::FWH;
    PushEvalC4<BOX> E;
    E.init(P, l);

    for (int r = 0; r * kPushGroups < n_live; r++) {
        const int idx = r * kPushGroups + grp;
        const bool have = idx < n_live;
        if (!__any(have)) break;
        const int ptid = (int)list[have ? idx : 0];
        const int oly = ptid >> 4, olx = 2 * (ptid & 15) + ((oly + colour) & 1);
        const int npx = x0 + olx, npy = y0 + oly;
        const float4 pl = norm4[PM_AT(P, npy * cols + npx, np, kChkNorm4)];
        const int tnx = olx + halo, tny = oly + halo;
        const float nxf = (float)npx, nyf = (float)npy;
        E.nyf = nyf;
        const int cpx = npx + cdx, cpy = npy + cdy;
        const bool cvalid = have && cpx >= 0 && cpx < cols && cpy >= 0 && cpy < rows;
        const int tpx = tnx + cdx, tpy = tny + cdy;
        const float4 centre = *reinterpret_cast<const float4 *>(lds + LY::tile_a + 4 * (tpy * twc + (tpx >> 1)));
        const float *ta_c = lds + LY::tile_a + 4 * ((tpy - R) * twc + ((tpx - R) >> 1));

        ViewCombiner<true> comb;
        float Hl[9], H[9];
        for (int v = 0; v < n; v++) {
            if ((v & (kPushLanes - 1)) == 0) homography(P->rc.K_inv, P->view[min(v + l, n - 1)], pl, Hl);
#pragma unroll
            for (int k = 0; k < 9; k++) H[k] = __shfl(Hl[k], grp_lane0 + (v & (kPushLanes - 1)));
            const float reach = (float)(R + kPushReach);
            const bool fast = __all(window_z_safe(H, nxf - reach, nxf + reach, nyf - reach, nyf + reach));
            const gptr_bytes packed = (gptr_bytes)P->view[v].packed.raw;
            if (fast) {
                E.template family<true, N, LY::FWH, R, R + kPushReach>(H, packed, lds, dgrp, nxf, tnx, tny);
                E.template family<true, LY::FWH, N, R + kPushReach, R>(H, packed, lds, dgrp + LY::hbase, nxf, tnx, tny);
            } else {
                E.template family<false, N, LY::FWH, R, R + kPushReach>(H, packed, lds, dgrp, nxf, tnx, tny);
                E.template family<false, LY::FWH, N, R + kPushReach, R>(H, packed, lds, dgrp + LY::hbase, nxf, tnx, tny);
            }
            __builtin_amdgcn_wave_barrier();
            const float c = push_chain_c4<BOX>(ta_c, centre, lds, dgrp + dbase, jstride);
            __builtin_amdgcn_wave_barrier();
            comb.add(c, v, nullptr);
        }
        const float F = comb.finish(P, n, nullptr);
        if (cvalid) P->push_cost[PM_AT(P, (size_t)l * np + (size_t)(cpy * cols + cpx), 8 * np, kChkPushCost)] = F;
    }
}

}  // namespace pm
