// pm_prop_shared.h -- spatial propagation of one checkerboard colour with SHARED patch samples.
//
// What it replaces: the propagation rounds of pm::sweep_kernel, i.e. the bodies of
// gipuma_checkerboard_spatialPropClose_cu / ...Far_cu (reference gipuma.cu:1471-1588, 1353-1468).
//
// Observation (exact, not an approximation).  The patch cost of plane pi at pixel p in view v is
//     c_v(p, pi) = sum over the window samples q = p + (i, j) of  w(p, q) * dis_v(q, pi)
// accumulated by fmaf in the order i outer, j inner (gipuma.cu:633-676).  dis_v(q, pi) -- the warp
// of q through the homography of (pi, v), the five bilinear taps, the truncated colour/gradient
// differences (gipuma.cu:207-274) -- does NOT depend on p: the homography is a function of the plane
// and the view only, q enters as exact small integers.  Only the support weight w(p, q) and the
// summation belong to p.  Spatial propagation hands the SAME plane (bit for bit, gipuma.cu:847-850)
// to many pixels: a pixel whose plane changed offers it to its 8 checkerboard neighbours, and the
// pixels that took it offer it on.  On config C a workgroup tile holds, in the late half-sweeps,
// ~540 surviving (pixel, plane) tasks but only ~90 distinct planes, and the windows of the pixels
// that test one plane overlap: the distinct samples are 27-50 % of the per-task samples
// (scripts/exp/et_stats.py).  So:
//
//   1. tasks of the tile (after the exact skipping rules of sweep_read_state) are grouped by the
//      bits of their plane (hash table in LDS) and by the parity class of their pixel (window
//      offsets are even, so pixels of different parity never share a sample): a SUBGROUP;
//   2. phase A: for every subgroup, dis_v is evaluated ONCE on the bounding box of its pixels'
//      windows -- one sample per lane, a wavefront taking 64 consecutive points (row by row) of
//      one subgroup so that the homography sits in scalar registers -- and left in LDS;
//   3. phase B: one lane per task runs the reference's 64-term fmaf chain over its own window,
//      reading dis from LDS and its support weights from registers (computed once per task, used
//      for every view).  Same terms, same order, same roundings as view_cost_pipe: bit-identical.
//   4. the owner lanes then replay the accepts in the reference order (sweep_replay).
//
// Per view the two phases are separated by workgroup barriers; subgroups are processed in batches
// that fit the LDS sample buffer.  The refinement stage (unique planes, nothing to share) stays in
// pm::sweep_kernel, launched right after with stages = REFINE.
//
// Supported: gray window-packed planes with float-encoded offsets, compile-time box 11 or 15,
// best-N combination with n_best <= 4 (register combiner).  Everything else takes the old path.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int kPsReach = 5;                              // propagation distance (gipuma.cu:1437-1462)
constexpr int kPsExtW = kTileW + 2 * kPsReach;           // 42: source cells per row of the extended tile
constexpr int kPsExtH = kSweepTileH + 2 * kPsReach;      // 26
constexpr int kPsCells = kPsExtW * kPsExtH;              // 1092, half of them of the other colour
constexpr int kPsBatchSg = 64;                           // subgroups per batch (LDS records, double-buffered)
constexpr int kPsRec = 16;                               // floats per subgroup record
constexpr int kPsHash = 1024;                            // hash slots (<= 546 keys)
constexpr int kPsLdsBudget = 20224;                      // 32-bit words: 79 KiB -> two workgroups per CU

template <int BOX>
struct PsLayout {  // offsets in 32-bit words into the dynamic LDS array
    static constexpr int R = (BOX - 1) / 2, N = R + 1;   // window radius; samples per window side
    static constexpr int hw = (BOX + 1) / 2;             // tile halo (Win::halo_w)
    static constexpr int tw = kTileW + 2 * hw, th = kSweepTileH + 2 * hw;
    static constexpr int tile = kLutSize;
    static constexpr int bres = tile + 4 * tw * th;      // [8][256] task costs (SweepLane::bres)
    static constexpr int btask = bres + 8 * kThreads;    // 2048 u16 tasks, sorted by subgroup
    static constexpr int cellgid = btask + 1024;         // 1092 u16: source cell -> group id
    static constexpr int sgmask = cellgid + 546;         // 1092 u32: x mask | y mask << 16 of a subgroup's pixels
    static constexpr int tcnt = sgmask + 1092;           // 546 u32 task counters (two classes); later the batch list
    static constexpr int tstart = tcnt + 548;            // 1093 u16: first task of a subgroup
    static constexpr int cp = tstart + 548;              // 1093 u32: first sample point (padded to 64) of a subgroup
    static constexpr int grpsrc = cp + 1096;             // 546 u32: a pixel that holds the group's plane
    static constexpr int misc = grpsrc + 546;            // counters
    static constexpr int sgmap = misc + 16;              // D/64 bytes: wavefront item -> subgroup of the batch
    static constexpr int htab = sgmap + 32;              // [2][kPsBatchSg][kPsRec] per-view subgroup records
    static constexpr int dis = htab + 2 * kPsBatchSg * kPsRec;
    static constexpr int D = ((kPsLdsBudget - dis) / 64) * 64;  // sample points per batch (64-point wavefront items)
    static constexpr int total = dis + D;
    static constexpr int maxpts = (15 + N) * (7 + N);    // largest bounding box of a subgroup
    static_assert(D >= kPsHash && D >= ((maxpts + 63) & ~63) && D <= 32 * 4 * 64, "sample buffer");
    static_assert(tw * th <= 8 * kThreads + 1024, "staging plane of stage_tile must end before live data");
    static_assert(D < 8192 && maxpts < 512, "record packing");
    static_assert(total - bres >= TpLayout::total, "the refinement stage reuses the space from bres on");
};

struct PsReq {  // one sample point in flight
    float a, b;      // bilinear fractions
    u32x4_a4 w;      // 4x4 texel window
    int taddr;       // byte offset of the reference texel in the LDS tile
    int slot;        // index into the sample buffer, -1 = padding
};

__device__ __forceinline__ int ps_cell(int k, int lx, int ly)
{
    const int dist = k < 4 ? 1 : kPsReach;
    const int sx = lx + ((k & 3) == 2 ? -dist : (k & 3) == 3 ? dist : 0) + kPsReach;
    const int sy = ly + ((k & 3) == 0 ? -dist : (k & 3) == 1 ? dist : 0) + kPsReach;
    return sy * kPsExtW + sx;
}

// exclusive prefix sum of `v` over the 256 lanes of the workgroup; *total = sum.  `wc` = 4 ints of LDS.
__device__ __forceinline__ int ps_scan256(int v, int *wc, int *total)
{
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += up;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // previous users of wc are done
    if ((threadIdx.x & 63) == 63) wc[wave] = incl;
    __syncthreads();
    const int c0 = wc[0], c1 = wc[1], c2 = wc[2], c3 = wc[3];
    *total = c0 + c1 + c2 + c3;
    return incl - v + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
}

// experiment builds only (-DPM_PS_PROFILE): cycles along wavefront 0's path, summed over workgroups
#ifdef PM_PS_PROFILE
#define PS_PROF_DECL unsigned long long ps_acc[12] = {0}; unsigned long long ps_last = wall_clock64();
#define PS_T(i)                                          \
    do {                                                 \
        if (threadIdx.x == 0) {                          \
            const unsigned long long now_ = wall_clock64(); \
            ps_acc[i] += now_ - ps_last;                 \
            ps_last = now_;                              \
        }                                                \
    } while (0)
#define PS_PROF_FLUSH                                                                     \
    do {                                                                                  \
        if (threadIdx.x == 0 && P->prof)                                                  \
            for (int i_ = 0; i_ < 12; i_++) atomicAdd(&P->prof[i_], ps_acc[i_]);           \
    } while (0)
#else
#define PS_PROF_DECL
#define PS_T(i)
#define PS_PROF_FLUSH
#endif

template <int BOX>
__global__ __launch_bounds__(kThreads, 2) void prop_shared_kernel(const Problem *__restrict__ P,
                                                                  float4 *__restrict__ norm4,
                                                                  float *__restrict__ cost, int colour,
                                                                  unsigned stages, unsigned tune, uint32_t phase)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = PsLayout<BOX>;
    constexpr int N = LY::N, R = LY::R, tw = LY::tw, hw = LY::hw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = P->cols;
    const int n_sel = P->n_sel;

    PS_PROF_DECL
    SweepLane L;
    sweep_read_state<BOX, 1>(L, P, lds, norm4, cost, colour, stages & 3u, tune, true);
    const unsigned needmask = L.needmask;
    PS_T(0);

    unsigned short *btask = L.btask;
    unsigned short *cellgid = reinterpret_cast<unsigned short *>(lds + LY::cellgid);
    unsigned *sgmask = reinterpret_cast<unsigned *>(lds + LY::sgmask);
    unsigned *tcnt = reinterpret_cast<unsigned *>(lds + LY::tcnt);
    unsigned short *tstart = reinterpret_cast<unsigned short *>(lds + LY::tstart);
    unsigned *cp = reinterpret_cast<unsigned *>(lds + LY::cp);
    unsigned *grpsrc = reinterpret_cast<unsigned *>(lds + LY::grpsrc);
    int *misc = reinterpret_cast<int *>(lds + LY::misc);
    unsigned char *sgmap = reinterpret_cast<unsigned char *>(lds + LY::sgmap);
    float *htab = lds + LY::htab;
    float *dis = lds + LY::dis;
    unsigned *hash = reinterpret_cast<unsigned *>(dis);  // only while the groups are formed

    // ---- 1. groups: distinct planes among the source pixels that have a surviving task ----
    for (int c = tid; c < 546; c += kThreads) reinterpret_cast<unsigned *>(cellgid)[c] = 0xffffffffu;
    for (int c = tid; c < 1092; c += kThreads) sgmask[c] = 0u;
    for (int c = tid; c < 548; c += kThreads) tcnt[c] = 0u;
    for (int c = tid; c < kPsHash; c += kThreads) hash[c] = 0xffffffffu;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++)
        if ((needmask >> k) & 1u) cellgid[ps_cell(k, L.lx, L.ly)] = 0xfffe;  // source cell in use
    __syncthreads();
    constexpr int kCellsPerLane = (kPsCells + kThreads - 1) / kThreads;
    int leader[kCellsPerLane];
    int nlead = 0;
#pragma unroll
    for (int e = 0; e < kCellsPerLane; e++) {
        const int c = tid + kThreads * e;
        leader[e] = -1;
        if (c < kPsCells && cellgid[c] == 0xfffe) {
            const int gi = (L.y0 + c / kPsExtW - kPsReach) * cols + (L.x0 + c % kPsExtW - kPsReach);
            const float4 pl = norm4[gi];
            unsigned h = mix32(__float_as_uint(pl.x) ^ mix32(__float_as_uint(pl.y) ^
                               mix32(__float_as_uint(pl.z) ^ mix32(__float_as_uint(pl.w))))) & (kPsHash - 1);
            for (;;) {
                const unsigned old = atomicCAS(&hash[h], 0xffffffffu, (unsigned)c);
                if (old == 0xffffffffu) {
                    leader[e] = c;
                    break;
                }
                const int go = (L.y0 + (int)old / kPsExtW - kPsReach) * cols + (L.x0 + (int)old % kPsExtW - kPsReach);
                if (same_bits(pl, norm4[go])) {
                    leader[e] = (int)old;
                    break;
                }
                h = (h + 1) & (kPsHash - 1);
            }
            if (leader[e] == c) nlead++;
        }
    }
    int n_groups;
    int g = ps_scan256(nlead, misc, &n_groups);
#pragma unroll
    for (int e = 0; e < kCellsPerLane; e++) {
        const int c = tid + kThreads * e;
        if (leader[e] == c) {
            cellgid[c] = (unsigned short)g;
            grpsrc[g] = (unsigned)((L.y0 + c / kPsExtW - kPsReach) * cols + (L.x0 + c % kPsExtW - kPsReach));
            g++;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kCellsPerLane; e++) {
        const int c = tid + kThreads * e;
        if (leader[e] >= 0 && leader[e] != c) cellgid[c] = cellgid[leader[e]];
    }
    __syncthreads();  // also: the hash table (aliasing the sample buffer) is dead from here on
    PS_T(1);

    // ---- 2. subgroups (group, parity class of the pixel): pixel masks, task counts and ranks ----
    const int Xt = L.lx >> 1, Yt = L.ly >> 1, cls = L.lx & 1;
    unsigned rk0 = 0, rk1 = 0;  // rank of this lane's tasks inside their subgroup, 8 bits per slot
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((needmask >> k) & 1u) {
            const int gid = cellgid[ps_cell(k, L.lx, L.ly)];
            atomicOr(&sgmask[2 * gid + cls], (1u << Xt) | (1u << (16 + Yt)));
            const unsigned old = atomicAdd(&tcnt[gid], cls ? 0x10000u : 1u);
            const unsigned rank = cls ? old >> 16 : old & 0xffffu;
            if (k < 4)
                rk0 |= rank << (8 * k);
            else
                rk1 |= rank << (8 * (k - 4));
        }
    }
    __syncthreads();
    // exclusive prefix sums over the subgroups: tasks and (padded) sample points
    const int n_sg = 2 * n_groups;
    constexpr int kSgPerLane = 5;  // 1280 >= 1092 + 1
    {
        int tc[kSgPerLane], np[kSgPerLane], tsum = 0, psum = 0;
#pragma unroll
        for (int i = 0; i < kSgPerLane; i++) {
            const int sg = tid * kSgPerLane + i;
            tc[i] = 0;
            np[i] = 0;
            if (sg < n_sg) {
                const unsigned m = sgmask[sg];
                tc[i] = (int)((tcnt[sg >> 1] >> (16 * (sg & 1))) & 0xffffu);
                if (m) {
                    const unsigned xm = m & 0xffffu, ym = m >> 16;
                    const int W = (31 - __clz((int)xm)) - (__ffs((int)xm) - 1) + N;
                    const int Hh = (31 - __clz((int)ym)) - (__ffs((int)ym) - 1) + N;
                    np[i] = (W * Hh + 63) & ~63;
                }
            }
            tsum += tc[i];
            psum += np[i];
        }
        int ttot, ptot;
        int tex = ps_scan256(tsum, misc, &ttot);
        int pex = ps_scan256(psum, misc, &ptot);
#pragma unroll
        for (int i = 0; i < kSgPerLane; i++) {
            const int sg = tid * kSgPerLane + i;
            if (sg <= n_sg) {
                tstart[sg] = (unsigned short)tex;
                cp[sg] = (unsigned)pex;
            }
            tex += tc[i];
            pex += np[i];
        }
    }
    __syncthreads();
    // tasks sorted by subgroup
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if ((needmask >> k) & 1u) {
            const int gid = cellgid[ps_cell(k, L.lx, L.ly)];
            const unsigned rank = ((k < 4 ? rk0 >> (8 * k) : rk1 >> (8 * (k - 4)))) & 0xffu;
            btask[tstart[2 * gid + cls] + rank] = (unsigned short)(tid | (k << 8));
        }
    }
    // batches: runs of subgroups whose samples fit the buffer and whose tasks fit one lane each
    unsigned short *blist = reinterpret_cast<unsigned short *>(tcnt);  // the counters are dead (barrier above)
    if (wave == 0) {
        int s = 0, nb = 0;
        while (s < n_sg) {
            int cnt = 0;
            for (int rnd = 0; rnd < 2; rnd++) {
                const int e = s + 1 + 64 * rnd + lane;
                const bool ok = e <= n_sg && (e - s) <= kPsBatchSg && (int)(cp[min(e, n_sg)] - cp[s]) <= LY::D &&
                                (int)tstart[min(e, n_sg)] - (int)tstart[s] <= kThreads;
                const int c = __popcll(__ballot(ok));  // ok is monotone in e: a prefix of ones
                cnt += c;
                if (c < 64) break;
            }
            if (lane == 0) blist[nb] = (unsigned short)s;
            nb++;
            s += cnt;  // >= 1: a single subgroup always fits (static_asserts of PsLayout, <= 128 tasks)
        }
        if (lane == 0) {
            blist[nb] = (unsigned short)n_sg;
            misc[8] = nb;
        }
    }
    __syncthreads();
    const int n_batches = misc[8];
    PS_T(2);

    // ---- 3. batches ----
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const char *lut_magic = (const char *)lds - kMagicBits;
    const char *tile_bytes = (const char *)(lds + LY::tile);
    for (int b = 0; b < n_batches; b++) {
        const int s_lo = blist[b], s_hi = blist[b + 1];
        const int t_lo = tstart[s_lo], n_tb = (int)tstart[s_hi] - t_lo;
        const unsigned p_lo = cp[s_lo];
        const int npts = (int)(cp[s_hi] - p_lo);
        const int n_sgb = s_hi - s_lo;
        if (n_tb == 0) continue;  // (subgroups without tasks have no points either)

        // this lane's task: support weights of its window (weight_cu, gipuma.cu:186-193) and where its
        // window starts in the sample buffer
        const bool has_task = tid < n_tb;
        int owner = 0, slot = 0, dofs = 0, W_t = N;
        float w[N * N];
#pragma unroll
        for (int q = 0; q < N * N; q++) w[q] = 0.0f;
        if (has_task) {
            const unsigned code = btask[t_lo + tid];
            owner = (int)(code & 255u);
            slot = (int)(code >> 8);
            int olx, oly;
            owner_pixel(L, owner, colour, olx, oly);
            const int sg = 2 * (int)cellgid[ps_cell(slot, olx, oly)] + (olx & 1);
            const unsigned m = sgmask[sg];
            const unsigned xm = m & 0xffffu, ym = m >> 16;
            const int X0 = __ffs((int)xm) - 1, Y0 = __ffs((int)ym) - 1;
            W_t = (31 - __clz((int)xm)) - X0 + N;
            dofs = (int)(cp[sg] - p_lo) + ((oly >> 1) - Y0) * W_t + ((olx >> 1) - X0);
            const float *tp0 = L.tile + ((oly + hw) * tw + (olx + hw)) * 4;
            const float centre = tp0[0];
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int j = 0; j < N; j++) {
                    const float colorDis = __builtin_fabsf(tp0[4 * ((2 * j - R) * tw + (2 * i - R))] - centre);
                    w[i * N + j] = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
                }
        }
        PS_T(3);
        ViewCombiner<true> comb;
        // The lanes at the top of the workgroup (tasks fill it from the bottom) keep one subgroup of the
        // batch each -- its plane and geometry in registers -- and write its per-view record
        // (homography, reciprocal safety, lattice origin) one view ahead of phase A, into the other half
        // of the double-buffered table, while the task lanes run phase B.
        const int rl = kThreads - 1 - tid;
        bool rec_live = false;
        float4 rpl = make_float4(0.f, 0.f, -1.f, 1.f);
        float rbx = 0.f, rby = 0.f, rqx1 = 0.f, rqy1 = 0.f, rWf = 8.f, rtb = 0.f, rpk = 0.f;
        if (rl < n_sgb) {
            const int sg = s_lo + rl;
            const unsigned m = sgmask[sg];
            if (m) {
                rec_live = true;
                const unsigned xm = m & 0xffffu, ym = m >> 16;
                const int X0 = __ffs((int)xm) - 1, Y0 = __ffs((int)ym) - 1;
                const int W = (31 - __clz((int)xm)) - X0 + N, Hh = (31 - __clz((int)ym)) - Y0 + N;
                const int scls = sg & 1;
                const int lx0 = 2 * X0 + scls - R, ly0 = 2 * Y0 + ((colour + scls) & 1) - R;  // tile-relative
                rbx = (float)(L.x0 + lx0);
                rby = (float)(L.y0 + ly0);
                rqx1 = rbx + (float)(2 * (W - 1));
                rqy1 = rby + (float)(2 * (Hh - 1));
                rWf = (float)W;
                rtb = (float)((ly0 + hw) * tw + (lx0 + hw));
                const unsigned pbase = cp[sg] - p_lo;
                rpk = __uint_as_float(pbase | ((unsigned)(W * Hh) << 16));
                rpl = norm4[grpsrc[sg >> 1]];
                const unsigned q1 = (pbase + (((unsigned)(W * Hh) + 63u) & ~63u)) >> 6;
                for (unsigned q = pbase >> 6; q < q1; q++) sgmap[q] = (unsigned char)rl;  // item -> subgroup
            }
        }
        auto write_record = [&](int v, float *tab) {
            if (rec_live) {
                float H[9];
                homography(P->rc.K_inv, P->view[v], rpl, H);
                const bool safe = window_z_safe(H, rbx, rqx1, rby, rqy1);
                float *rec = tab + rl * kPsRec;
                *reinterpret_cast<float4 *>(rec) = make_float4(H[0], H[1], H[2], H[3]);
                *reinterpret_cast<float4 *>(rec + 4) = make_float4(H[4], H[5], H[6], H[7]);
                *reinterpret_cast<float4 *>(rec + 8) = make_float4(H[8], safe ? 1.0f : 0.0f, rbx, rby);
                *reinterpret_cast<float4 *>(rec + 12) = make_float4(rWf, 1.0f / rWf, rtb, rpk);
            }
        };
        if (n_sel > 0) write_record(0, htab);
        PS_T(4);
        const int n_items = npts >> 6;
        const int T = (n_items - wave + 3) >> 2;  // this wavefront's items: wave, wave + 4, ...
        for (int v = 0; v < n_sel; v++) {
            const float *hcur = htab + (v & 1) * (kPsBatchSg * kPsRec);
            float *hnext = htab + ((v + 1) & 1) * (kPsBatchSg * kPsRec);
            __syncthreads();  // records of view v visible; phase B of the previous view is done with dis
            PS_T(5);
            // --- phase A: dis of every sample point of the batch.  A wavefront takes 64 consecutive
            //     points = one chunk of ONE subgroup, so the subgroup's record is wave-uniform: 16 lanes
            //     fetch it and readlane moves it to scalar registers.  Software pipeline over the
            //     wavefront's items, one stage per iteration, so that every LDS / global access has a
            //     whole iteration to complete (two wavefronts per SIMD cannot hide them otherwise):
            //       A  item t     subgroup index (LDS byte)
            //       B  item t-1   record (LDS)
            //       C  item t-2   point coordinates, warp, window request (global)
            //       C' item t-3   reference texel request (LDS)
            //       D  item t-4   taps, dis, store
            const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)P->view[v].packed - (uintptr_t)kMagicBits);
            {
                PsReq qa, qb;  // two window requests in flight: items t-2 and t-3 (t-4 is being reduced)
                qa.a = qa.b = qb.a = qb.b = 0.0f;
                qa.w = qb.w = u32x4_a4{0u, 0u, 0u, 0u};
                qa.taddr = qb.taddr = 0;
                qa.slot = qb.slot = -1;
                float4 t4a = make_float4(0.f, 0.f, 0.f, 0.f), t4b = t4a;
                int ja = 0, jb = 0, rva = 0, rvb = 0;
                // one pipeline step; the caller alternates the register sets so that nothing has to be
                // copied (a copy of a register with a load in flight would wait for the load)
                // (every memory access of the steady-state loop is issued unconditionally, so that the
                //  compiler's s_waitcnt counts are exact -- a window request under a branch makes it wait
                //  for ALL outstanding loads at the next use)
                auto step = [&](int t, const bool doD, const bool doC, PsReq &qcur, const PsReq &qoth,
                                const float4 &t4in, float4 &t4out, int jin, int &jout, int rvin, int &rvout) {
                    // A
                    jout = (int)sgmap[min(wave + 4 * t, n_items - 1)];
                    // B (subgroup index read in the previous step)
                    rvout = __float_as_int(hcur[__builtin_amdgcn_readfirstlane(jin) * kPsRec + (lane & 15)]);
                    // D: item t-4, requested two steps ago into qcur, its reference texel in t4in
                    if (doD) {
                        const Taps tp5 = taps_u8(qcur.a, qcur.b, qcur.w.x, qcur.w.y, qcur.w.z, qcur.w.w);
                        // pmCostComputation_shared, gipuma.cu:251-274
                        const float colDiff = t4in.w - tp5.sc;
                        const float gradX = t4in.y - tp5.gx2;
                        const float gradY = t4in.z - tp5.gy2;
                        const float gradDis =
                            min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
                        const float colDis = min_abs_nc(colDiff, tau_color);
                        const float d = __builtin_fmaf(alpha, gradDis, oma * colDis);
                        if (qcur.slot >= 0) dis[qcur.slot] = d;
                    }
                    // C: item t-2 (record read in the previous step); items past the end repeat the last one
                    if (doC) {
                        const int it = min(wave + 4 * (t - 2), n_items - 1);
                        float r[kPsRec];
#pragma unroll
                        for (int q = 0; q < kPsRec; q++) r[q] = __int_as_float(__builtin_amdgcn_readlane(rvin, q));
                        const unsigned pk = __float_as_uint(r[15]);
                        const int k = it * 64 + lane - (int)(pk & 0xffffu), n = (int)(pk >> 16);
                        const float kf = (float)min(k, n - 1);
                        // the points of a subgroup run row by row (k = Y * W + X): neighbouring lanes sample
                        // neighbouring texels of the same image rows and LDS banks of the reference tile
                        const float Yf = __builtin_floorf((kf + 0.5f) * r[13]);  // k / W, exact (k < 512, W < 32)
                        const float Xf = __builtin_fmaf(-Yf, r[12], kf);
                        const float qx = __builtin_fmaf(2.0f, Xf, r[10]), qy = __builtin_fmaf(2.0f, Yf, r[11]);
                        // getCorrespondingPoint_cu, gipuma.cu:207-217, same fmaf nesting as view_cost_pipe
                        const float X = __builtin_fmaf(r[1], qy, __builtin_fmaf(r[0], qx, r[2]));
                        const float Y = __builtin_fmaf(r[4], qy, __builtin_fmaf(r[3], qx, r[5]));
                        const float Z = __builtin_fmaf(r[7], qy, __builtin_fmaf(r[6], qx, r[8]));
                        // IEEE 1/Z by the exact fast path where the record says the subgroup's box is safe
                        const float rz = r[9] != 0.0f ? rcp_newton(Z) : 1.0f / Z;
                        const float sx = X * rz, sy = Y * rz;
                        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
                        qcur.a = sx - fx0;
                        qcur.b = sy - fy0;
                        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
                        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
                        const uint32_t off = __float_as_uint(__builtin_fmaf(Yc, pwf, Xc + magic_c));
                        qcur.w = *(gptr_u32x4)(magic_base + off);
                        qcur.taddr = (int)__builtin_fmaf(Yf, (float)(2 * tw), __builtin_fmaf(Xf, 2.0f, r[14])) * 16;
                        qcur.slot = (t - 2 < T && k < n) ? it * 64 + lane : -1;
                    }
                    // C': reference texel of item t-3 (requested in the previous step into qoth)
                    t4out = *reinterpret_cast<const float4 *>(tile_bytes + qoth.taddr);
                };
                // fill: steps 0..3 (two windows in flight afterwards)
                step(0, false, false, qa, qb, t4a, t4b, ja, jb, rva, rvb);
                step(1, false, false, qb, qa, t4b, t4a, jb, ja, rvb, rva);
                step(2, false, true, qa, qb, t4a, t4b, ja, jb, rva, rvb);
                step(3, false, true, qb, qa, t4b, t4a, jb, ja, rvb, rva);
                // steady state: steps 4 .. T+1, in pairs (an odd count is padded with one repeated item)
                const int t_end = 4 + ((max(T - 2, 0) + 1) & ~1);
                for (int t = 4; t < t_end; t += 2) {
                    step(t, true, true, qa, qb, t4a, t4b, ja, jb, rva, rvb);
                    __builtin_amdgcn_sched_barrier(0);
                    step(t + 1, true, true, qb, qa, t4b, t4a, jb, ja, rvb, rva);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // drain: the two windows still in flight
                step(t_end, true, false, qa, qb, t4a, t4b, ja, jb, rva, rvb);
                step(t_end + 1, true, false, qb, qa, t4b, t4a, jb, ja, rvb, rva);
            }
            PS_T(6);
            __syncthreads();
            PS_T(7);
            // --- phase B: the reference's summation (columns outer, rows inner, one fmaf per sample)
            if (has_task) {
                const float *dc = dis + dofs;
                float c = 0.0f;
#pragma unroll
                for (int i = 0; i < N; i++)
#pragma unroll
                    for (int j = 0; j < N; j++) c = __builtin_fmaf(w[i * N + j], dc[j * W_t + i], c);
                comb.add(c, v, nullptr);
            }
            PS_T(8);
            if (v + 1 < n_sel) write_record(v + 1, hnext);
            PS_T(9);
        }
        if (has_task) L.bres[slot * kThreads + owner] = comb.finish(P, n_sel, nullptr);
        // (no barrier: the next batch's records are not read by phase B, and its samples are written
        //  only after the barrier that follows its records)
    }
    __syncthreads();

    // ---- 4. accepts in the reference order, write back (gipuma.cu:1585-1587) ----
    PS_T(10);
    sweep_replay(L, P, norm4);
    // ---- 5. refinement in the same launch (stages & 4): the steps of pm::sweep_kernel, by (candidate,
    //         view) items where that pays (refine_step_items), else unbounded ----
    if (stages & 4u) {
        RefineDraws Rd;
        refine_init(Rd, P, stages);
        const bool et_on = P->et_enable && !(tune & Tune::kNoEarlyExit);
        if (et_on && blockIdx.x == 0 && tid < (int)kEtSlot) P->et_stat[((phase + 1u) % 3u) * kEtSlot + tid] = 0u;
        refine_begin(Rd, L, P, phase);
        if (tid < 2) reinterpret_cast<int *>(L.bres + TpLayout::cnt)[tid] = 0;
        __syncthreads();  // the replay has read its candidate costs; refine_two_phase reuses that space
        int tp_seq = 0;
        const Win<BOX> win(P);
        for (int step = 0; step < Rd.nref; step++) {
            float d_new = 0.f;
            float4 cand = make_float4(0.f, 0.f, -1.f, 1.f);
            const bool do_eval = L.active;
            if (do_eval) cand = refine_candidate(Rd, L, P, d_new);
            refine_next_step(Rd);
            if (et_on && step < 3 && !(tune & Tune::kNoTwoPhase) &&
                refine_step_items<BOX, 1>(P, L, lds, colour, phase, step, do_eval, cand, d_new, tp_seq))
                continue;
            if (do_eval) {
                const float *tp0 = L.tile + ((L.ly + L.hh) * L.tw + (L.lx + L.hw)) * 4;
                const float c = multiview_cost<BOX, true, true, true, 1>(P, tp0, L.tw, lds, L.cv, L.px, L.py, cand, win);
                if (c < L.cst) {  // refinement has no depth-range test, :986
                    L.depth = d_new;
                    L.pl = cand;
                    L.cst = c;
                    L.chg = 1;
                }
            }
        }
    }
    if (L.active) {
        cost[L.center] = L.cst;
        norm4[L.center] = L.pl;
        P->changed[L.center] = (unsigned char)L.chg;
    }
    PS_T(11);
    PS_PROF_FLUSH;
}

}  // namespace pm
