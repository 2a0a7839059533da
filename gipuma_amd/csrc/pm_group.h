// pm_group.h -- propagation costs of a half-sweep evaluated once per PLANE, not once per (pixel, plane).
//
// What it replaces: the cost evaluations inside gipuma_checkerboard_spatialPropClose_cu / ...Far_cu
// (reference gipuma.cu:1471-1588, 1353-1468; pmCostMultiview_cu :720-806 at :865-872) of the half-sweep that
// follows, for the half-sweeps in which pm::push_kernel (pm_push.h) no longer pays.  The accept tests stay with
// the consumer (sweep_replay in pm_sweep.h, Tune::kPushConsume), exactly as with pm::push_kernel.
//
// Observation (exact, as in pm_push.h).  The patch cost of plane pi at pixel p in view v is
//     c_v(p, pi) = sum over the window samples q = p + (2i-R, 2j-R) of  w(p, q) * dis_v(q, pi)
// accumulated by fmaf, i outer, j inner (gipuma.cu:633-676), and dis_v(q, pi) does not depend on p.  After the
// first half-sweeps a plane that fits a surface patch has spread over it: the SAME plane, bit for bit, is the
// candidate of many pixels of a tile at once (it is held by several neighbours, each offering it to its up to
// eight consumers), and the windows of those pixels overlap.  On config C the candidates a tile still has to
// evaluate (after the skip rules (A), (D), (H) of sweep_kernel) fall into groups of on average 5 with one
// plane and one sample lattice, whose windows cover 0.31-0.38 of the samples the tasks have one by one
// (scripts/exp/et_stats.py, profiles/r02_exp_sharing_stats_cpu.txt).
//
// So, per tile of the colour about to be swept:
//   tasks   the (pixel, candidate slot) pairs that must be evaluated, exactly as sweep_setup finds them;
//   groups  tasks with bitwise equal planes and the same window lattice (window offsets are odd: the samples of
//           a pixel have the other x parity) -- a hash table in LDS, then a counting sort by group;
//   strips  a group's samples fill the bounding box of its windows on that lattice: (8 + spread/2) columns of
//           (8 + spread/2) samples.  One lane per column ("strip") walks it with the instruction sequence of
//           view_cost_pipe (same X0/Y0/Z0 per column, same fmaf per row: the same bits) and leaves dis in LDS;
//   chain   one lane per task then runs the reference's 64-term fmaf chain over ITS window -- its own support
//           weights, the reference's order -- and feeds the view cost to its ViewCombiner.
// Round 3's version looped super-batches of groups with two workgroup barriers per view and lost to the fused
// kernel (its strips, 8-15 samples long and of mixed length inside a wavefront, ran at a third of the fused loop's
// issue rate; its chain phase was bound by three dependent LDS reads per term).  This version has NO workgroup
// barrier after the set-up:
//   * the groups are sorted by strip length, so the strips a wavefront walks together are equally long;
//   * a BATCH -- at most kGrpBatchGroups groups, 64 strips, 64 tasks and as many samples per view as a wavefront's
//     slice of LDS holds: groups from the long-strip end of the order until one of these is full, then groups from
//     the short-strip end into the lanes still free -- belongs to ONE wavefront, which takes it from a shared
//     two-ended cursor; strips and chains of a batch only meet inside that wavefront (LDS operations of a wavefront
//     complete in order);
//   * per view a lane walks its strips in one continuous software pipeline (the windows of sample s+2 requested
//     before sample s is reduced, across strip boundaries), and the first two windows of the NEXT view are requested
//     before the chain phase of this one, so their latency hides behind it;
//   * the support weights of a task depend on its pixel alone: they are computed once per batch and stay in
//     registers, one lane per task (boxes 11, 15, colour: at most 64 floats); a chain term is one LDS read and one
//     fmaf.  Box 25 has 169: batches of up to 32 tasks run two lanes per task (the first sums window columns 0..6 and
//     hands its partial sum to the second), larger ones one lane per task with the weights as byte indices into the
//     weight table, four to a register, read a window column ahead;
//   * a strip's samples lie an odd number of words apart in the wavefront's buffer (LDS banks);
//   * the homographies of a batch's (group, view) pairs are computed four views at a time, one pair per lane;
//   * 50 KB of LDS and at most 168 registers: three workgroups per CU (box 25 and colour: 256 registers, two).
// The aggregate goes to Problem::push_cost[slot][pixel], where the half-sweep finds it (Tune::kPushConsume);
// candidates the skip rules removed get MAXCOST there, which the strict < of the accept test (gipuma.cu:868)
// rejects like their true cost would be.
// Same terms, same order, same roundings as view_cost_pipe + multiview_cost: bit-identical.
//
// Supported: window-packed planes (gray: float-encoded offsets), best-N with n_best <= 4; gray boxes 11 / 15 / 25 alone
// (group_kernel) or fused with the half-sweep (sweep_group_kernel), colour box 15 alone.
// GIPUMA_HIP_COUNTS=1 reports batch statistics and phase clocks.
#pragma once
#include <type_traits>
#include "pm_device.h"

namespace pm {

constexpr int kGrpMaxTasks = 8 * kThreads;  // 2048
constexpr int kGrpHashSize = 2048;

constexpr int kGrpBatchStrips = 64;    // one strip per lane
// Lanes per task in the chains.  One lane holds the support weights of its window columns in registers, as floats: 64 for
// box 15, 36 for box 11.  Box 25 has 169: two lanes per task, the first sums window columns 0..6 (91 weights) and hands
// its partial sum to the second, which continues with columns 7..12 -- the reference's order.  (One lane with the weights
// as byte indices into the table, four per register, costs a table read, a bit-field extract and an address per term:
// 3 VALU + 1.5 LDS instructions instead of 1 + 0.5, on 31 % of the lanes: the form batches of more than 32 tasks still
// use, see group_costs.  -DPM_GROUP_TASK_LANES25=1 builds it for every batch.)
#ifndef PM_GROUP_TASK_LANES
#define PM_GROUP_TASK_LANES 1
#endif
#ifndef PM_GROUP_TASK_LANES25
#define PM_GROUP_TASK_LANES25 2
#endif
template <int BOX>
__host__ __device__ constexpr int group_task_lanes()
{
    return (BOX == 25 || BOX == 19) ? PM_GROUP_TASK_LANES25 : PM_GROUP_TASK_LANES;  // (box 19: 100 weights, 50 per lane)
}
constexpr int kGrpBatchGroups = 8;     // (a group has at least N strips: 6 / 8 / 13 for boxes 11 / 15 / 25)
// dis values of one view a wavefront's LDS slice holds: 64 strips of 16 (boxes 11, 15: three workgroups per CU) or of
// 20 (box 25: two workgroups per CU) rows
template <int BOX>
__host__ __device__ constexpr int group_batch_samples()
{
    return BOX == 25 ? 1280 : 1024;
}
constexpr int kGrpViewsPerH = 4;       // homographies are prepared this many views at a time (8 groups x 4 views = 32 lanes;
                                       // 8 views: 6 KB more LDS per workgroup, which the sample buffers use better)
#ifndef PM_GROUP_LAPS
#define PM_GROUP_LAPS 0x7f  // which of the fused kernel's phase clocks are compiled in (bit = slot)
#endif
#ifndef PM_GROUP_WG
#define PM_GROUP_WG 3  // workgroups per CU the kernels are compiled for (3: 168 VGPRs)
#endif
#ifndef PM_GROUP_WG25
#define PM_GROUP_WG25 2  // ... box 25 (91 chain weights per lane, 13 samples per chain column) and colour: 256 registers
#endif
template <int BOX, int CH = 1>
__host__ __device__ constexpr int group_wg()
{
    return (BOX == 25 || CH == 4) ? PM_GROUP_WG25 : PM_GROUP_WG;  // (colour: a 24 KB float4 tile, three windows per sample)
}

template <int BOX, int CH = 1>
struct GroupLayout {  // offsets in 32-bit words into the dynamic LDS array
    static_assert(BOX == 11 || BOX == 15 || BOX == 19 || BOX == 25, "instantiated window sizes");
    static_assert(CH == 1 || BOX == 15, "colour: box 15");
    static constexpr int R = (BOX - 1) / 2, N = R + 1;
    static constexpr int task_lanes = group_task_lanes<BOX>();
    static_assert(task_lanes == 1 || task_lanes == 2, "one lane per task, or two that split its window columns");
    static constexpr int batch_tasks = 64;                 // tasks of a batch (two lanes per task: chains of 32 at a time, or one lane each)
    static constexpr int NH = (N + task_lanes - 1) / task_lanes;  // window columns of a chain lane (the second of two: N - NH)
    // the support weights of a chain lane's window columns stay in registers: as floats while they are at most 64,
    // else (box 25: 169) as their table indices |dI|, four per register -- a chain term then costs a table read more
    static constexpr bool byte_weights = NH * N > 96;
    static constexpr int tw = kTileW + 2 * N, th = kSweepTileH + 2 * N;
    static constexpr int max_rows = N + (kSweepTileH - 1) / 2;  // samples per strip: 8 + 7 = 15 for box 15
    static constexpr int max_cols = N + (kTileW - 1) / 2;       // strips per group: 8 + 15 = 23
    // sweep_read_state<.., PLANE_ONLY> stages the scalar plane of reference texels -- clamp-to-edge point samples --
    // right behind the weight table; the gradients are two subtractions per sample, the ones stage_tile does.
    // Colour: the float4 {B, G, R, 0} tile of the sweep kernels.
    static constexpr int plane = lut_size<CH>();            // [th][tw] (x CH)
    static constexpr int misc = plane + CH * tw * th;       // counters
    static constexpr int meta = misc + 128;                 // [2048] per group: first task | min lx << 11 | min ly << 16 | (ncols - N) << 20 | (nrows - N) << 24
    static constexpr int sbt = meta + kGrpMaxTasks + 4;     // [2048] u16: the tasks (owner | slot << 8) ordered by group
    static constexpr int gorder = sbt + kGrpMaxTasks / 2;   // [2048] u16: group ids ordered by strip length
    // per wavefront: the tables, homographies and samples of the batch it is working on
    static constexpr int w_tab = 0;                                   // [kGrpBatchGroups][4] u16: group id, first strip, first task
    static constexpr int w_sgroup = w_tab + 2 * kGrpBatchGroups;      // [kGrpBatchStrips] u8: group (in the batch) of a strip
    static constexpr int w_tgroup = w_sgroup + kGrpBatchStrips / 4;   // [64] u8: ... of a task
    static constexpr int w_gplane = w_tgroup + 16;                    // [kGrpBatchGroups] float4  (64 bytes for tgroup)
    static constexpr int w_hbuf = w_gplane + 4 * kGrpBatchGroups;     // [kGrpViewsPerH][kGrpBatchGroups][12]: H, fast flag
    static constexpr int batch_samples = group_batch_samples<BOX>();
    static constexpr int w_dis = w_hbuf + kGrpViewsPerH * kGrpBatchGroups * 12;  // [batch_samples]
    static constexpr int w_stride = w_dis + batch_samples;
    static constexpr int waves = gorder + kGrpMaxTasks / 2;  // [4] of the above
    // while grouping, the wavefronts' areas hold: hash table | group id of a task (u16) | task list (u16) | counters
    static constexpr int g_hash = waves, g_gid = g_hash + kGrpHashSize, g_btask = g_gid + kGrpMaxTasks / 2,
                         g_cnt = g_btask + kGrpMaxTasks / 2;
    static constexpr int total = waves + (4 * w_stride > g_cnt + kGrpMaxTasks - waves ? 4 * w_stride : g_cnt + kGrpMaxTasks - waves);
    static_assert(batch_samples >= max_cols * (max_rows | 1), "the largest possible group fits a wavefront's sample buffer");
    static_assert(kGrpBatchStrips >= max_cols, "the strips of the largest possible group fit the lanes");
    static_assert(total * 4 * group_wg<BOX, CH>() <= 160 * 1024, "workgroups per CU");
    // Every strip of a batch walks the BATCH's row count (GroupWalk::body, nr_b = the longest group's), so a strip of a
    // shorter group near the bottom of the tile reads reference texels of up to 2 (max_rows - N) rows (+ 1 for the lower
    // neighbour of the central difference) below the staged plane.  Those words are the misc / meta / sbt tables that
    // follow `plane`: the reads stay inside this workgroup's allocation, the dis values computed from them are never read
    // by a chain (no task of that group has those rows).  A layout change must keep the over-read inside the allocation:
    static_assert(plane + CH * tw * (th + 2 * (max_rows - N) + 1) <= total, "the strips' over-read below the plane stays inside the allocation");
    static_assert((w_gplane % 4) == 0 && (w_hbuf % 4) == 0 && (waves % 4) == 0 && (w_stride % 4) == 0, "16-byte aligned float4 tables");
};

// The 100 MHz clock for the phase counters.  s_memrealtime returns through the scalar memory path, whose results come back
// out of order: while one is outstanding every LDS wait the compiler inserts must be lgkmcnt(0) instead of a graded count,
// and a clock value that is only needed at the NEXT lap stays outstanding through whatever lies in between (measured: the
// box-25 chains, 169 table reads + 169 sample reads per view, each waited for on its own -- config D 17 % slower with
// one lap anywhere in the kernel).  The empty asm consumes the value on the spot.
__device__ __forceinline__ unsigned long long lap_clock()
{
    unsigned long long t = wall_clock64();
    asm volatile("" : "+s"(t));
    return t;
}

__device__ __forceinline__ uint32_t plane_hash(float4 pl, int cls)
{
    uint32_t h = __float_as_uint(pl.x) * 0x9E3779B1u;
    h = (h ^ __float_as_uint(pl.y)) * 0x85EBCA77u;
    h = (h ^ __float_as_uint(pl.z)) * 0xC2B2AE3Du;
    h = (h ^ __float_as_uint(pl.w)) * 0x27D4EB2Fu;
    h ^= h >> 15;
    return (h + (uint32_t)cls) & (kGrpHashSize - 1);
}

// One strip of one view for the walk below: the per-column terms of view_cost_pipe (warp_col: getCorrespondingPoint_cu,
// gipuma.cu:207-217) and the row coefficients of the view's homography
struct StripView {
    WarpCol wc;
    WarpRow wr;
};

// The strip of one lane for one view.  Per sample the arithmetic is view_cost_pipe's (getCorrespondingPoint_cu
// :207-217, the five bilinear taps :251-253, pmCostComputation_shared :254-274) without the weight and the
// accumulation; {I, gx1, gy1} of the reference texel are formed from the scalar plane as stage_tile forms them
// (gipuma.cu:254-259: central differences).
//
// The rows of the strip are dealt alternately to two register sets.  A set's step: convert the twelve texels of
// the window it holds, THEN request the window of its next row (two rows further) into the registers just freed,
// then reduce the converted texels.  Every load so has the rest of its own step and the other set's whole step to
// arrive, and the loop needs no register rotation although its trip count is not a compile-time constant.  Every
// path through the walk has a fixed number of loads in flight, so the waits the compiler inserts are exact.
template <int BOX>
struct GroupWalk {
    using LY = GroupLayout<BOX>;
    static constexpr int tw = LY::tw;
    MagicAddr MA;  // wave-uniform
    DisConst K;    // (alpha / 16 and 16 tau_g: dis_fold, pm_sample.h)
    const Problem *P;
    // this lane's strip: first sample point, reference texel of that point in the plane, where its dis values go,
    // and its group's slot in the batch tables
    float qx, qy0;
    const float *tcol;
    float *out;
    int grp;

    struct Set {  // one of the two register sets
        float qy;          // sample row of its next request
        const float *tq;   // reference texel of the row it holds, biased by -(tw + 1) texels: the texel and its four neighbours
                           // then lie at non-negative constant offsets, which go into the LDS instructions (an LDS offset
                           // is unsigned: tq[-1], tq[-tw] cost an address addition each, per sample)
        float *o;          // where that row's dis goes
        WinReq req;        // the window it holds (or that is on its way)
    };

    __device__ __forceinline__ void init(const Problem *__restrict__ P_)
    {
        P = P_;
        MA = magic_addr(P_);
        K = dis_const(P_);
    }
    __device__ __forceinline__ StripView view_of(const float *__restrict__ hb) const
    {
        // hb: [kGrpBatchGroups][12] of the view; three 16-byte reads
        const float4 a = *reinterpret_cast<const float4 *>(hb + 12 * grp);
        const float4 b = *reinterpret_cast<const float4 *>(hb + 12 * grp + 4);
        const float4 c = *reinterpret_cast<const float4 *>(hb + 12 * grp + 8);
        const float H[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
        StripView s;
        s.wc = warp_col(H, qx);
        s.wr = warp_row(H);
        return s;
    }
    // (`fast`, wave-uniform: the cheap division is exact on the batch's boxes for this view.  ONE loop with a scalar branch
    //  per sample instead of two specialised loops: with two, the waits the compiler placed in the second one drained
    //  both window loads at the top of every iteration -- vmcnt(0) where the first has vmcnt(1) twice --, which left a
    //  load only the other set's reduce to arrive in.)
    __device__ __forceinline__ WinReq request(gptr_bytes magic_base, const StripView &s, float qy, bool fast) const
    {
        return magic_request_rt(MA, magic_base, s.wc, s.wr, qy, fast);
    }
    // window words w0..w3 = columns X..X+3, byte r = row Y+r
    static __device__ __forceinline__ Tex12 unpack(const u32x4_a4 &w) { return unpack12(w.x, w.y, w.z, w.w); }
    // (I: the reference texel of the sample; xr, xl, yd, yu: its right / left / lower / upper neighbours)
    __device__ __forceinline__ void reduce(gptr_bytes magic_base, const Tex12 &t, float a, float b, float I, float xr, float xl,
                                           float yd, float yu, float *__restrict__ o) const
    {
        const float gx1 = xr - xl;
        const float gy1 = yd - yu;
        const Taps tp5 = sample_taps_gray(a, b, t, plane_of_magic(magic_base, P));
        const float colDiff = I - tp5.sc;
        const float gradX = gx1 - tp5.gx2;
        const float gradY = gy1 - tp5.gy2;
        *o = dis_folded<true>(__builtin_fabsf(gradX) + __builtin_fabsf(gradY), colDiff, K.alpha16, K.oma, K.tau_color, K.taug16);
    }
    // the first two window requests of a view (rows 0 and 1)
    __device__ __forceinline__ void first(gptr_bytes magic_base, const float *__restrict__ hb, StripView &sv, Set &A,
                                          Set &B, bool fast) const
    {
        sv = view_of(hb);
        A.req = request(magic_base, sv, qy0, fast);
        B.req = request(magic_base, sv, qy0 + 2.0f, fast);
        A.qy = qy0 + 4.0f;
        B.qy = qy0 + 6.0f;
        A.tq = tcol - (tw + 1);
        B.tq = tcol + 2 * tw - (tw + 1);
        A.o = out;
        B.o = out + 1;
    }
    // A step of a set: convert the window it holds, request its next one (two rows further) unless LAST, reduce.
    template <bool LAST>
    __device__ __forceinline__ void step(gptr_bytes magic_base, const StripView &sv, Set &S, bool fast) const
    {
        const float *tq = S.tq;
        float *o = S.o;
        // (the reference texels first: their LDS latency passes during the conversions and the request)
        const float I = tq[tw + 1], xr = tq[tw + 2], xl = tq[tw], yd = tq[2 * tw + 1], yu = tq[1];
        const Tex12 t = unpack(S.req.w);
        const float a = S.req.a, b = S.req.b;
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            S.req = request(magic_base, sv, S.qy, fast);
            S.qy += 4.0f;
            S.tq = tq + 4 * tw;
            S.o = o + 2;
        }
        __builtin_amdgcn_sched_barrier(0);
        reduce(magic_base, t, a, b, I, xr, xl, yd, yu, o);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the nr rows (wave-uniform, >= 3) of this lane's strip for the view whose first two requests are in (A, B):
    // A walks the even rows, B the odd ones; nothing is in flight at the end
    __device__ __forceinline__ void body(gptr_bytes magic_base, int nr, const StripView &sv, Set &A, Set &B, bool fast) const
    {
        int fi = __builtin_amdgcn_readfirstlane(fast ? 1 : 0);
        const int n_pairs = (nr - 2) >> 1;  // pairs of rows both of which request (rows r, r + 1 with r + 3 < nr)
        for (int r = 0; r < n_pairs; r++) {
#if !PM_APPROX  // (the approx flavour has one reciprocal: nothing to unswitch)
            asm volatile("" : "+s"(fi));  // (opaque per iteration: the loop is not to be unswitched into two)
#endif
            step<false>(magic_base, sv, A, fi != 0);
            step<false>(magic_base, sv, B, fi != 0);
        }
        if (nr & 1) {  // (wave-uniform) rows nr - 3 (requests the last one), nr - 2, nr - 1
            step<false>(magic_base, sv, A, fi != 0);
            step<true>(magic_base, sv, B, false);
            step<true>(magic_base, sv, A, false);
        } else {
            step<true>(magic_base, sv, A, false);
            step<true>(magic_base, sv, B, false);
        }
    }
};

// The same walk for -color_processing (T = float4; view_cost_c4_loop in pm_cost.h): three 16-byte window loads and
// three tap sets per sample (integer window addressing: three words per texel do not fit the float-encoded offsets),
// l1_norm(float4) reductions in the reference's order (gipuma.cu:174-179), the reference-side terms from the float4
// {B, G, R, 0} tile: the texel and its four neighbours.
template <int BOX>
struct GroupWalkC4 {
    using LY = GroupLayout<BOX, 4>;
    static constexpr int tw = LY::tw;
    DisConst K;  // wave-uniform (alpha / 16 and 16 tau_g: dis_fold, pm_sample.h)
    IntAddr IA;
    const Problem *P;
    float qx, qy0;
    const float *tcol;  // the float4 texel of the strip's first sample point
    float *out;
    int grp;

    typedef WinReq3 Req;
    struct Set {
        float qy;
        const float *tq;
        float *o;
        Req req;
    };

    __device__ __forceinline__ void init(const Problem *__restrict__ P_)
    {
        P = P_;
        K = dis_const(P_);
        IA = int_addr(P_);
    }
    __device__ __forceinline__ StripView view_of(const float *__restrict__ hb) const
    {
        const float4 a = *reinterpret_cast<const float4 *>(hb + 12 * grp);
        const float4 b = *reinterpret_cast<const float4 *>(hb + 12 * grp + 4);
        const float4 c = *reinterpret_cast<const float4 *>(hb + 12 * grp + 8);
        const float H[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
        StripView s;
        s.wc = warp_col(H, qx);
        s.wr = warp_row(H);
        return s;
    }
    __device__ __forceinline__ Req request(gptr_bytes packed, const StripView &s, float qy, bool fast) const
    {
        return c4_request_rt(IA, packed, s.wc, s.wr, qy, fast);
    }
    __device__ __forceinline__ void first(gptr_bytes packed, const float *__restrict__ hb, StripView &sv, Set &A, Set &B,
                                          bool fast) const
    {
        sv = view_of(hb);
        A.req = request(packed, sv, qy0, fast);
        B.req = request(packed, sv, qy0 + 2.0f, fast);
        A.qy = qy0 + 4.0f;
        B.qy = qy0 + 6.0f;
        A.tq = tcol - 4 * (tw + 1);  // (biased like GroupWalk::Set::tq)
        B.tq = tcol + 4 * 2 * tw - 4 * (tw + 1);
        A.o = out;
        B.o = out + 1;
    }
    template <bool LAST>
    __device__ __forceinline__ void step(gptr_bytes packed, const StripView &sv, Set &S, bool fast) const
    {
        const float *tq = S.tq;
        float *o = S.o;
        const float4 lv = *reinterpret_cast<const float4 *>(tq + 4 * (tw + 1));
        const float4 left = *reinterpret_cast<const float4 *>(tq + 4 * tw), right = *reinterpret_cast<const float4 *>(tq + 4 * (tw + 2));
        const float4 up = *reinterpret_cast<const float4 *>(tq + 4), down = *reinterpret_cast<const float4 *>(tq + 4 * (2 * tw + 1));
        // word 3k+c = column k, channel c (view_cost_c4_loop)
        const Req &q = S.req;
        const Tex12 tb = unpack12(q.q0.x, q.q0.w, q.q1.z, q.q2.y);
        const Tex12 tg = unpack12(q.q0.y, q.q1.x, q.q1.w, q.q2.z);
        const Tex12 tr = unpack12(q.q0.z, q.q1.y, q.q2.x, q.q2.w);
        const float a = q.a, b = q.b;
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            S.req = request(packed, sv, S.qy, fast);
            S.qy += 4.0f;
            S.tq = tq + 4 * 4 * tw;
            S.o = o + 2;
        }
        __builtin_amdgcn_sched_barrier(0);
        Taps t[3];
        sample_taps_c4(a, b, tb, tg, tr, plane_of(packed, P), t);
        const Taps &t0 = t[0], &t1 = t[1], &t2 = t[2];
        // pmCostComputation_shared for T = float4, gipuma.cu:251-274
        const float colDiff = l1_3(lv.x - t0.sc, lv.y - t1.sc, lv.z - t2.sc);
        const float gX = l1_3((right.x - left.x) - t0.gx2, (right.y - left.y) - t1.gx2, (right.z - left.z) - t2.gx2);
        const float gY = l1_3((down.x - up.x) - t0.gy2, (down.y - up.y) - t1.gy2, (down.z - up.z) - t2.gy2);
        *o = dis_folded<false>(gX + gY, colDiff, K.alpha16, K.oma, K.tau_color, K.taug16);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void body(gptr_bytes packed, int nr, const StripView &sv, Set &A, Set &B, bool fast) const
    {
        int fi = __builtin_amdgcn_readfirstlane(fast ? 1 : 0);
        const int n_pairs = (nr - 2) >> 1;
        for (int r = 0; r < n_pairs; r++) {
#if !PM_APPROX
            asm volatile("" : "+s"(fi));
#endif
            step<false>(packed, sv, A, fi != 0);
            step<false>(packed, sv, B, fi != 0);
        }
        if (nr & 1) {
            step<false>(packed, sv, A, fi != 0);
            step<true>(packed, sv, B, false);
            step<true>(packed, sv, A, false);
        } else {
            step<true>(packed, sv, A, false);
            step<true>(packed, sv, B, false);
        }
    }
};

// The propagation costs of the tile whose state sweep_read_state<BOX, 1, 0, true> has just put into `L` (tile plane
// at lds + kLutSize, L.needmask = the candidate slots that must be evaluated): Problem::push_cost[slot][pixel] for
// every such slot.  All 256 lanes of the workgroup call it; it ends without a barrier (the wavefronts leave their
// batch loops one by one).  `write_skipped`: slots the consumer will replay although a skip rule removed them get
// MAXCOST (the stand-alone kernel: its consumer replays by rule (H) alone).
template <int BOX, int CH = 1>
__device__ __forceinline__ void group_costs(const Problem *__restrict__ P, SweepLane &L, float *lds,
                                            const float4 *__restrict__ norm4, int colour, int hist, bool write_skipped)
{
    using LY = GroupLayout<BOX, CH>;
    constexpr int R = LY::R, N = LY::N, tw = LY::tw, NH = LY::NH, TL = LY::task_lanes, kBatchTasks = LY::batch_tasks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = P->rows, cols = P->cols, n = P->n_sel;
    const size_t np = (size_t)rows * (size_t)cols;

    // (GIPUMA_HIP_COUNTS: 100 MHz wall-clock ticks per phase, summed over the workgroups, in row 62 of Problem::dbg)
    // (wave-uniform, so that the clock stays in scalar registers)
    const bool prof = P->dbg != nullptr && __builtin_amdgcn_readfirstlane(tid) == 0;
    unsigned long long tick = prof ? lap_clock() : 0ull;
    auto lap = [&](int slot) {
        if (prof) {
            const unsigned long long now = lap_clock();
            if (lane == 0) dbg_add(&P->dbg[62 * kDbgSlots + slot], now - tick);
            tick = now;
        }
    };
    unsigned short *sbt = reinterpret_cast<unsigned short *>(lds + LY::sbt);
    unsigned short *gorder = reinterpret_cast<unsigned short *>(lds + LY::gorder);
    uint32_t *meta = reinterpret_cast<uint32_t *>(lds + LY::meta);
    int *misc = reinterpret_cast<int *>(lds + LY::misc);
    uint32_t *hash = reinterpret_cast<uint32_t *>(lds + LY::g_hash);
    unsigned short *gid_of = reinterpret_cast<unsigned short *>(lds + LY::g_gid);
    unsigned short *btask = reinterpret_cast<unsigned short *>(lds + LY::g_btask);
    uint32_t *gcnt = reinterpret_cast<uint32_t *>(lds + LY::g_cnt);
    const float *plane = lds + LY::plane;
    const char *lut_magic = (const char *)lds - kMagicBits;

    // every slot the consumer will replay gets a cost: MAXCOST where a skip rule says "cannot be accepted"
    if (write_skipped && L.active) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int nb;
            if (!neighbour(k, L.px, L.py, rows, cols, L.center, nb)) continue;
            const bool replayed = !hist || P->changed[PM_AT(P, nb, np, kChkFlags)] != 0;
            if (replayed && !((L.needmask >> k) & 1u)) P->push_cost[PM_AT(P, (size_t)k * np + (size_t)L.center, 8 * np, kChkPushCost)] = kMaxCost;
        }
    }

    // ---- task list, owner-major (the order cannot matter: a task is a pure function of (pixel, plane)) ----
    int n_tasks;
    {
        const int cnt = __popc(L.needmask);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) misc[wave] = incl;
        for (int k = tid; k < kGrpHashSize; k += kThreads) hash[k] = 0u;
        if (tid == 0) {
            misc[8] = 0;   // number of groups
            misc[9] = 0;   // batch cursor: groups taken from the short-strip end | from the long-strip end << 16
        }
        if (tid < 16) misc[32 + tid] = 0;  // histogram of the groups' strip lengths
        __syncthreads();
        const int c0 = misc[0], c1 = misc[1], c2 = misc[2], c3 = misc[3];
        int pos = incl - cnt + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        n_tasks = c0 + c1 + c2 + c3;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((L.needmask >> k) & 1u) btask[pos++] = (unsigned short)(tid | (k << 8));
        __syncthreads();
    }
    lap(1);  // MAXCOST writes + task list
    if (n_tasks == 0) return;  // (uniform)

    // plane of a task descriptor (owner | slot << 8): the plane of the owner's neighbour in that slot
    auto desc_plane = [&](unsigned bt, int &olx, int &oly, int &nb) -> float4 {
        owner_pixel(L, (int)(bt & 255u), colour, olx, oly);
        const int epx = L.x0 + olx, epy = L.y0 + oly;
        neighbour((int)(bt >> 8), epx, epy, rows, cols, epy * cols + epx, nb);
        return norm4[PM_AT(P, nb, np, kChkNorm4)];
    };

    // ---- groups: tasks with bitwise equal planes and the same sample lattice ----
    // pass 1: the first task to claim a hash slot represents its group; gid_of[t] = representative's task index
    for (int t = tid; t < n_tasks; t += kThreads) {
        int olx, oly, nb;
        const float4 pl = desc_plane(btask[t], olx, oly, nb);
        const int cls = olx & 1;
        uint32_t h = plane_hash(pl, cls);
        int rep = t;
        for (;;) {
            const uint32_t seen = atomicCAS(&hash[h], 0u, (uint32_t)t + 1u);
            if (seen == 0u) break;  // claimed: this task represents a new group
            const int r = (int)seen - 1;
            int rlx, rly, rnb;
            const float4 rpl = desc_plane(btask[r], rlx, rly, rnb);
            if ((rlx & 1) == cls && (rnb == nb || same_bits(rpl, pl))) {
                rep = r;
                break;
            }
            h = (h + 1u) & (kGrpHashSize - 1);
        }
        gid_of[t] = (unsigned short)rep;
    }
    __syncthreads();
    // pass 2: dense group ids for the representatives (the hash table's memory now maps representative -> id)
    for (int t = tid; t < n_tasks; t += kThreads)
        if (gid_of[t] == (unsigned short)t) {
            const int g = atomicAdd(&misc[8], 1);
            hash[t] = (uint32_t)g;
            gcnt[g] = 0u;  // member count, then fill position
        }
    __syncthreads();
    const int n_groups = misc[8];
    // pass 3: every task learns its group id; member counts
    for (int t = tid; t < n_tasks; t += kThreads) {
        const int g = (int)hash[gid_of[t]];
        gid_of[t] = (unsigned short)g;
        atomicAdd(&gcnt[g], 1u);
    }
    __syncthreads();
    // exclusive prefix sum of the member counts (8 groups per lane): meta[g] = first task of group g
    {
        constexpr int per = kGrpMaxTasks / kThreads;
        uint32_t loc[per];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < per; e++) {
            const int g = tid * per + e;
            loc[e] = g < n_groups ? gcnt[g] : 0u;
            sum += (int)loc[e];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) misc[wave] = incl;
        __syncthreads();
        int run = incl - sum + (wave > 0 ? misc[0] : 0) + (wave > 1 ? misc[1] : 0) + (wave > 2 ? misc[2] : 0);
#pragma unroll
        for (int e = 0; e < per; e++) {
            const int g = tid * per + e;
            if (g < n_groups) {
                meta[g] = (uint32_t)run;
                gcnt[g] = 0u;  // fill counter of the scatter below
            }
            run += (int)loc[e];
        }
        __syncthreads();
    }
    // scatter: the task descriptors ordered by group
    for (int t = tid; t < n_tasks; t += kThreads) {
        const int g = gid_of[t];
        const int k = (int)atomicAdd(&gcnt[g], 1u);
        sbt[(int)meta[g] + k] = btask[t];
    }
    __syncthreads();
    // bounding box of a group's pixels, one lane per group; the histogram of the strip lengths
    for (int g = tid; g < n_groups; g += kThreads) {
        const int t0 = (int)meta[g], cnt = (int)gcnt[g];
        int mnx = 255, mxx = 0, mny = 255, mxy = 0;
        for (int k = 0; k < cnt; k++) {
            int olx, oly;
            owner_pixel(L, (int)(sbt[t0 + k] & 255u), colour, olx, oly);
            mnx = min(mnx, olx);
            mxx = max(mxx, olx);
            mny = min(mny, oly);
            mxy = max(mxy, oly);
        }
        meta[g] = (uint32_t)t0 | ((uint32_t)mnx << 11) | ((uint32_t)mny << 16) | ((uint32_t)((mxx - mnx) >> 1) << 20) |
                  ((uint32_t)((mxy - mny) >> 1) << 24);
        atomicAdd(&misc[32 + ((mxy - mny) >> 1)], 1);
    }
    __syncthreads();
    // groups ordered by strip length (counting sort, 8 possible lengths): the strips a wavefront walks together
    // are then equally long, and a batch's sample buffer is laid out with ONE row count
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b < 16; b++) {
            const int c = misc[32 + b];
            misc[32 + b] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int g = tid; g < n_groups; g += kThreads) {
        const int key = (int)((meta[g] >> 24) & 7u);
        gorder[atomicAdd(&misc[32 + key], 1)] = (unsigned short)g;
    }
    __syncthreads();  // LAST workgroup barrier: the grouping tables' memory becomes the wavefronts' batch areas
    lap(2);  // grouping, sort, bounding boxes, order
    if (P->dbg != nullptr && tid == 0) {
        dbg_add(&P->dbg[61 * kDbgSlots + 5], (unsigned long long)n_groups);
        dbg_add(&P->dbg[61 * kDbgSlots + 6], (unsigned long long)n_tasks);
        dbg_add(&P->dbg[61 * kDbgSlots + 7], 1ull);
    }
    // members of group g: sbt[first .. first + count)
    auto group_tasks = [&](int g, int &first) -> int {
        first = (int)(meta[g] & 2047u);
        return (g + 1 < n_groups ? (int)(meta[g + 1] & 2047u) : n_tasks) - first;
    };

    // ---- batches: each wavefront on its own ----
    float *wbase = lds + LY::waves + wave * LY::w_stride;
    unsigned short *wtab = reinterpret_cast<unsigned short *>(wbase + LY::w_tab);
    unsigned char *sgroup = reinterpret_cast<unsigned char *>(wbase + LY::w_sgroup);
    unsigned char *tgroup = reinterpret_cast<unsigned char *>(wbase + LY::w_tgroup);
    float4 *gplane = reinterpret_cast<float4 *>(wbase + LY::w_gplane);
    float *hbuf = wbase + LY::w_hbuf;
    float *dbuf = wbase + LY::w_dis;
    typename std::conditional<CH == 4, GroupWalkC4<BOX>, GroupWalk<BOX> >::type W;
    W.init(P);
    // Chains.  One lane per task with its N x N support weights in registers (boxes 11, 15, colour).  Box 25 (169 weights):
    // a batch of at most 32 tasks runs two lanes per task -- lane tl sums window columns 0..NH-1 of task tl from float
    // weights, lane 32 + tl continues with the rest --; a batch with more tasks runs one lane per task with the weights as
    // byte indices into the table, four per register (a table read, a bit-field extract and an address more per term).
    constexpr bool kDual = TL == 2;
    constexpr bool kHasFloat = kDual || !LY::byte_weights, kHasByte = kDual || LY::byte_weights;
    constexpr int kFloatW = kHasFloat ? NH * N : 0, kByteWords = kHasByte ? (N * N + 3) / 4 : 0;
    constexpr int kStore = kFloatW > kByteWords ? kFloatW : kByteWords;

    for (;;) {
        // -- take the next batch from the two-ended cursor: groups from the long-strip end of gorder until the lanes, the
        //    tasks or the sample buffer are full, then groups from the short-strip end into the lanes still free (they
        //    walk the batch's row count, a few rows more than their own: lanes that would otherwise idle) --
        int cur = 0;
        if (lane == 0) cur = __hip_atomic_load(&misc[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = __builtin_amdgcn_readfirstlane(cur);
        const int took_lo = cur & 0xffff, took_hi = cur >> 16;  // groups already taken from either end
        const int remaining = n_groups - took_lo - took_hi;
        if (remaining <= 0) break;
        // lanes 0..7: candidates from the top (longest strips first); lanes 8..15: from the bottom
        const bool top = lane < kGrpBatchGroups, bot = !top && lane < 2 * kGrpBatchGroups;
        const int rank = lane & (kGrpBatchGroups - 1);
        int g = -1, nc = 0, nt = 0, nr = 0, tfirst = 0;
        if ((top || bot) && rank < remaining) {
            g = (int)gorder[top ? n_groups - 1 - took_hi - rank : took_lo + rank];
            const uint32_t m = meta[g];
            nc = N + (int)((m >> 20) & 15u);
            nr = N + (int)((m >> 24) & 7u);
            nt = group_tasks(g, tfirst);
        }
        // inclusive prefix sums of strips and tasks inside each run of 8 lanes
        int ic = nc, it = nt;
#pragma unroll
        for (int d = 1; d < kGrpBatchGroups; d <<= 1) {
            const int uc = __shfl_up(ic, d), ut = __shfl_up(it, d);
            if (rank >= d) {
                ic += uc;
                it += ut;
            }
        }
        // (strip lengths do not increase from the top down: the first group's row count is the batch's; a strip's samples
        //  lie `nr | 1` words apart: an odd stride keeps the 64 lanes' writes of a row -- and the chains' reads -- off
        //  each other's LDS banks; an even one, 8 or 16 above all, serialises them)
        const int nr_b = __builtin_amdgcn_readfirstlane(nr);
        const int nrs = nr_b | 1;
        const bool fits_top = top && g >= 0 && ic <= kGrpBatchStrips && it <= kBatchTasks && ic * nrs <= LY::batch_samples;
        const unsigned ft = (unsigned)__ballot(fits_top) & 0xffu;
        int k1 = (int)__builtin_ctz(~ft);  // leading lanes that fit (<= 8)
        if (k1 == 0) k1 = 1;  // a single group always fits the strips and the buffer; more than 64 tasks: several rounds
        const int s1 = __builtin_amdgcn_readfirstlane(__shfl(ic, k1 - 1)), t1 = __builtin_amdgcn_readfirstlane(__shfl(it, k1 - 1));
        const bool fits_bot = bot && g >= 0 && rank < remaining - k1 && rank < kGrpBatchGroups - k1 && s1 + ic <= kGrpBatchStrips &&
                              t1 + it <= kBatchTasks && (s1 + ic) * nrs <= LY::batch_samples;
        const unsigned fbm = ((unsigned)(__ballot(fits_bot) >> kGrpBatchGroups)) & 0xffu;
        const int k2 = (int)__builtin_ctz(~fbm);
        const int cnt = k1 + k2;
        int won = 0;
        if (lane == 0) won = atomicCAS(&misc[9], cur, cur + k2 + (k1 << 16)) == cur ? 1 : 0;
        if (!__builtin_amdgcn_readfirstlane(won)) continue;  // another wavefront took from an end meanwhile
        // (wave-uniform, in scalar registers: they steer the loops below)
        const int n_strips = s1 + (k2 > 0 ? __builtin_amdgcn_readfirstlane(__shfl(ic, kGrpBatchGroups + k2 - 1)) : 0);
        const int n_alltasks = t1 + (k2 > 0 ? __builtin_amdgcn_readfirstlane(__shfl(it, kGrpBatchGroups + k2 - 1)) : 0);
        {
            // this lane's group becomes group j of the batch
            const int j = top ? (rank < k1 ? rank : -1) : bot ? (rank < k2 ? k1 + rank : -1) : -1;
            if (j >= 0) {
                const int sp = (top ? 0 : s1) + ic - nc, tp = (top ? 0 : t1) + it - nt;
                wtab[4 * j + 0] = (unsigned short)g;
                wtab[4 * j + 1] = (unsigned short)sp;
                wtab[4 * j + 2] = (unsigned short)tp;
                for (int k = 0; k < nc; k++) sgroup[sp + k] = (unsigned char)j;
                if (n_alltasks <= kBatchTasks)
                    for (int k = 0; k < nt; k++) tgroup[tp + k] = (unsigned char)j;
                int rlx, rly, rnb;
                gplane[j] = desc_plane(sbt[tfirst], rlx, rly, rnb);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (P->dbg != nullptr && lane == 0) {  // (GIPUMA_HIP_COUNTS: batch statistics in row 61 of Problem::dbg)
            unsigned long long *d = P->dbg + 61 * kDbgSlots;
            dbg_add(&d[0], 1ull);
            dbg_add(&d[1], (unsigned long long)n_strips);
            dbg_add(&d[2], (unsigned long long)n_alltasks);
            dbg_add(&d[3], (unsigned long long)cnt);
            dbg_add(&d[4], (unsigned long long)nr_b);
        }

        // -- this lane's strip (spare lanes shadow strip 0: the same values into the same places) --
        {
            const int s = lane < n_strips ? lane : 0;
            const int j = (int)sgroup[s];
            const uint32_t gm = meta[(int)wtab[4 * j]];
            const int c = s - (int)wtab[4 * j + 1];
            const int s_lx = (int)((gm >> 11) & 31u) - R + 2 * c, s_ly = (int)((gm >> 16) & 15u) - R;
            W.qx = (float)(L.x0 + s_lx);
            W.qy0 = (float)(L.y0 + s_ly);
            W.tcol = plane + CH * ((s_ly + L.hh) * tw + (s_lx + L.hw));
            W.out = dbuf + s * nrs;
            W.grp = j;
        }
        // homographies of (group, view) for views vb .. vb + kGrpViewsPerH - 1, one pair per lane; [9] = the fast
        // reciprocal is exact on the whole box (any such proof gives the bits of the IEEE division, see rcp_newton)
        auto h_block = [&](int vb) {
            const int j = lane & (kGrpBatchGroups - 1), vi = lane / kGrpBatchGroups, v = vb + vi;
            if (vi < kGrpViewsPerH && j < cnt && v < n) {
                const uint32_t gm = meta[(int)wtab[4 * j]];
                float H[9];
                homography(P->rc.K_inv, P->view[v], gplane[j], H);
                const int bx0 = L.x0 + (int)((gm >> 11) & 31u) - R, by0 = L.y0 + (int)((gm >> 16) & 15u) - R;
                const int bx1 = bx0 + 2 * (N - 1 + (int)((gm >> 20) & 15u)), by1 = by0 + 2 * (nr_b - 1);
                const bool safe = window_z_safe(H, (float)bx0, (float)bx1, (float)by0, (float)by1);
                float *h = hbuf + (vi * kGrpBatchGroups + j) * 12;
                *reinterpret_cast<float4 *>(h) = make_float4(H[0], H[1], H[2], H[3]);
                *reinterpret_cast<float4 *>(h + 4) = make_float4(H[4], H[5], H[6], H[7]);
                *reinterpret_cast<float4 *>(h + 8) = make_float4(H[8], safe ? 1.0f : 0.0f, 0.0f, 0.0f);
            }
        };
        auto view_fast = [&](const float *hb) -> bool { return __all(hb[12 * W.grp + 9] != 0.0f); };
        auto view_base = [&](int v) -> gptr_bytes {  // gray: the base of the float-encoded offsets; colour: the packed plane
            return CH == 4 ? (gptr_bytes)P->view[v].packed.raw : (gptr_bytes)((uintptr_t)P->view[v].packed.raw - (uintptr_t)kMagicBits);
        };

        // -- rounds of tasks (more than one only for a single group offered to more than 64 pixels) --
        const bool two = kDual && n_alltasks <= 32;          // (wave-uniform) two lanes per task, float weights
        const bool bytes = kHasByte && (!kHasFloat || !two);  // one lane per task, byte-indexed weights
        const int half = two ? lane >> 5 : 0, tl = two ? lane & 31 : lane;
        const int round_tasks = two ? 32 : 64;
        for (int t_lo = 0; t_lo < n_alltasks; t_lo += round_tasks) {
            const int n_bt = min(round_tasks, n_alltasks - t_lo);
            // this lane's task: sample-buffer offset of its (half) window, slot, pixel, support weights of its columns
            int t_off = 0, t_slot = 0, t_center = 0;
            const float *t_tp = plane + CH * (L.hh * tw + L.hw);
            const bool has = tl < n_bt;
            if (has) {
                const int j = n_alltasks <= kBatchTasks ? (int)tgroup[tl] : 0;
                const int gg = (int)wtab[4 * j];
                const uint32_t gm = meta[gg];
                const unsigned bt = sbt[(int)(gm & 2047u) + (t_lo + tl - (int)wtab[4 * j + 2])];
                int olx, oly;
                owner_pixel(L, (int)(bt & 255u), colour, olx, oly);
                t_off = ((int)wtab[4 * j + 1] + ((olx - (int)((gm >> 11) & 31u)) >> 1) + half * NH) * nrs +
                        ((oly - (int)((gm >> 16) & 15u)) >> 1);
                t_slot = (int)(bt >> 8);
                t_center = (L.y0 + oly) * cols + (L.x0 + olx);
                t_tp = plane + CH * ((oly + L.hh) * tw + (olx + L.hw));
            }
            // support weights (weight_cu, gipuma.cu:186-193: 256 possible weights) of this lane's window columns,
            // column outer, row inner -- the order of the chain; as floats, or as their table indices |dI| packed
            // four to a register (the registers are the same ones: a batch uses one form)
            float wst[kStore];
            if constexpr (CH == 4) {
                // colour: the table is indexed by the integer |dB| + |dG| + |dR| (0..765), view_cost_c4_loop
                const float4 centre = *reinterpret_cast<const float4 *>(t_tp);
                const float *tc = t_tp + 4 * (-R * tw - R);
#pragma unroll
                for (int i = 0; i < N; i++)
#pragma unroll
                    for (int jj = 0; jj < N; jj++) {
                        const float4 lv = *reinterpret_cast<const float4 *>(tc + 4 * (2 * jj * tw + 2 * i));
                        const float S = __builtin_fabsf(lv.x - centre.x) + __builtin_fabsf(lv.y - centre.y) +
                                        __builtin_fabsf(lv.z - centre.z);
                        wst[i * N + jj] = lds[(int)S];
                    }
            } else {
                const float centre = t_tp[0];
                if (!bytes) {
                    if constexpr (kHasFloat) {
                        const float *tc = t_tp + (-R * tw - R + 2 * half * NH);
#pragma unroll
                        for (int i = 0; i < NH; i++) {
                            // (two lanes, odd N: the second lane has one column less; its spare slot re-reads its first column)
                            const int ci = (kDual && N % 2 != 0 && i == NH - 1 && half != 0) ? 0 : i;
#pragma unroll
                            for (int jj = 0; jj < N; jj++) {
                                wst[i * N + jj] = lut_weight(lut_magic, tc[2 * jj * tw + 2 * ci], centre);  // |dI|: an integer 0..255
                            }
                        }
                    }
                } else {
                    if constexpr (kHasByte) {
                        const float *tc = t_tp + (-R * tw - R);
                        uint32_t word = 0u;
#pragma unroll
                        for (int e = 0; e < N * N; e++) {
                            const int i = e / N, jj = e - i * N;
                            const float colorDis = __builtin_fabsf(tc[2 * jj * tw + 2 * i] - centre);
                            word |= ((__float_as_uint(colorDis + kMagicF) >> 2) & 0xffu) << (8 * (e & 3));
                            if ((e & 3) == 3 || e == N * N - 1) {
                                wst[e >> 2] = __uint_as_float(word);
                                word = 0u;
                            }
                        }
                    }
                }
            }
            // the table entry of window sample e (byte form): byte e & 3 of its word, times four = its byte offset
            auto byte_weight = [&](int e) -> float {
                const uint32_t w = __float_as_uint(wst[e >> 2]);
                const uint32_t off = (e & 3) == 0 ? (w << 2) & 0x3fcu : (w >> (8 * (e & 3) - 2)) & 0x3fcu;
                return *(const float *)((const char *)lds + off);
            };
            ViewCombiner<true> comb;

            h_block(0);
            __builtin_amdgcn_wave_barrier();
            StripView sv;
            typename decltype(W)::Set SA, SB;
            bool fast = view_fast(hbuf);
            W.first(view_base(0), hbuf, sv, SA, SB, fast);
            for (int v = 0; v < n; v++) {
                // strips: dis of every sample column of the batch
                W.body(view_base(v), nr_b, sv, SA, SB, fast);
                __builtin_amdgcn_wave_barrier();
                // the next view's first windows travel during the chains
                if (v + 1 < n) {
                    if ((v + 1) % kGrpViewsPerH == 0) {
                        h_block(v + 1);
                        __builtin_amdgcn_wave_barrier();
                    }
                    const float *hn = hbuf + ((v + 1) % kGrpViewsPerH) * (kGrpBatchGroups * 12);
                    fast = view_fast(hn);
                    W.first(view_base(v + 1), hn, sv, SA, SB, fast);
                }
                // chains: the reference's summation order over each task's own window.  A window column's samples
                // (and, byte form, its weights) are read while the column before is summed.
                const float *dcol = dbuf + t_off;
                float cst = 0.0f;
                auto float_chain = [&](auto ncols) {  // this lane's first ncols window columns
                    constexpr int NC = decltype(ncols)::value;
                    float dv[2][N];
#pragma unroll
                    for (int jj = 0; jj < N; jj++) dv[0][jj] = dcol[jj];
#pragma unroll
                    for (int i = 0; i < NC; i++) {
                        if (i + 1 < NC) {
#pragma unroll
                            for (int jj = 0; jj < N; jj++) dv[(i + 1) & 1][jj] = dcol[(i + 1) * nrs + jj];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < N; jj++) cst = accum(wst[i * N + jj], dv[i & 1][jj], cst);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                auto byte_chain = [&]() {
                    // (the table reads depend on the pixel alone: left to itself the compiler hoists all of them out of
                    //  the view loop and spills; the empty asm makes the indices opaque per view)
#pragma unroll
                    for (int q = 0; q < kByteWords; q++) asm volatile("" : "+v"(wst[q]));
                    float dv[2][N], wv[2][N];
#pragma unroll
                    for (int jj = 0; jj < N; jj++) {
                        dv[0][jj] = dcol[jj];
                        wv[0][jj] = byte_weight(jj);
                    }
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        if (i + 1 < N) {
#pragma unroll
                            for (int jj = 0; jj < N; jj++) {
                                dv[(i + 1) & 1][jj] = dcol[(i + 1) * nrs + jj];
                                wv[(i + 1) & 1][jj] = byte_weight((i + 1) * N + jj);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < N; jj++) cst = accum(wv[i & 1][jj], dv[i & 1][jj], cst);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (bytes) {
                    if constexpr (kHasByte) {
                        if (has) {
                            byte_chain();
                            comb.add(cst, v, nullptr);
                        }
                    }
                } else if constexpr (kDual) {
                    if (half == 0 && has) float_chain(std::integral_constant<int, NH>());
                    const float left = __shfl(cst, tl);
                    if (half == 1 && has) {
                        cst = left;
                        float_chain(std::integral_constant<int, N - NH>());
                        comb.add(cst, v, nullptr);
                    }
                } else if constexpr (kHasFloat) {
                    if (has) {
                        float_chain(std::integral_constant<int, N>());
                        comb.add(cst, v, nullptr);
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the next view's strips overwrite the samples
            }
            if ((two ? half == 1 : true) && has) P->push_cost[PM_AT(P, (size_t)t_slot * np + (size_t)t_center, 8 * np, kChkPushCost)] = comb.finish(P, n, nullptr);
        }
    }
    lap(3);  // batches (this workgroup's first wavefront)
}

// grid = tiles of the sweep kernels; `colour`: the colour about to be swept (the consumers); `hist`: rule (H) is
// valid for that half-sweep (the consumer replays only the slots whose producer changed).  The stand-alone form:
// the half-sweep itself is a sweep launch with Tune::kPushConsume.
template <int BOX, int CH = 1>
__global__ __launch_bounds__(kThreads, (group_wg<BOX, CH>())) void group_kernel(const Problem *__restrict__ P,
                                                                              const float4 *__restrict__ norm4,
                                                                              const float *__restrict__ cost, int colour,
                                                                              int hist, unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    SweepLane L;
    // rule (H) comes in through `hist`, like in pm::push_kernel; the other rules exactly as the sweep applies them
    const unsigned t2 = (tune & ~(Tune::kPushConsume | Tune::kHistorySkip)) | (hist ? Tune::kHistorySkip : 0u);
    sweep_read_state<BOX, CH, 0, CH == 1>(L, P, lds, norm4, cost, colour, 3u, t2, true);
    group_costs<BOX, CH>(P, L, lds, norm4, colour, hist, true);
}

// One colour of one iteration (gipuma.cu:1353-1711, the three launches of sweep_kernel) with the propagation costs
// evaluated per PLANE: set-up and skip rules of sweep_kernel, group_costs, then sweep_kernel's own accept replay,
// refinement and write-back (sweep_body).  One launch instead of group_kernel + sweep_kernel: the tile and the state
// are read once, and the workgroups of a CU are in different stages at any time -- the refinement stage waits on
// scattered window loads, the strips of the propagation stage are bound by instruction issue.
// Window-packed planes (gray: float-encoded offsets), register combiner, box 11 / 15 / 25; colour: box 15.
template <int BOX, int CH = 1>
__global__ __launch_bounds__(kThreads, (group_wg<BOX, CH>())) void sweep_group_kernel(const Problem *__restrict__ P,
                                                                                    float4 *__restrict__ norm4,
                                                                                    float *__restrict__ cost, int colour,
                                                                                    uint32_t phase, unsigned tune)
{
#ifndef PM_FUSED_COLOUR_EXPERIMENT  // (builds of the miscompile hunt, DESIGN.md 9: scripts/gpu_r05_fusedcolour.sh)
    static_assert(CH == 1, "gray only: colour sessions run pm::group_kernel<15, 4> and the sweep kernel (gipuma_hip_create)");
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = GroupLayout<BOX, CH>;
    constexpr int tw = LY::tw, th = LY::th;
    SweepLane L;
#ifdef PM_WG_TICKS
    const unsigned long long wt0 = wall_clock64();
#endif
    // (GIPUMA_HIP_COUNTS: wall-clock ticks of the stages around group_costs, row 62 of Problem::dbg, slots 0, 4..6)
    const bool prof = PM_GROUP_LAPS != 0 && P->dbg != nullptr && __builtin_amdgcn_readfirstlane((int)threadIdx.x) == 0;
    unsigned long long tick = prof ? lap_clock() : 0ull;
    auto lap = [&](int slot) {
        if (!((PM_GROUP_LAPS >> slot) & 1)) return;
        if (prof) {
            const unsigned long long now = lap_clock();
            if ((threadIdx.x & 63) == 0) dbg_add(&P->dbg[62 * kDbgSlots + slot], now - tick);
            tick = now;
        }
    };
    const unsigned long long t_start = P->tile_clock != nullptr ? lap_clock() : 0ull;
    sweep_read_state<BOX, CH, 0, CH == 1>(L, P, lds, norm4, cost, colour, 7u, tune & ~Tune::kPushConsume, true,
                                          (const int *)P->tile_order.raw);
    lap(0);
    group_costs<BOX, CH>(P, L, lds, norm4, colour, (tune & Tune::kHistorySkip) != 0, false);
    if (PM_GROUP_LAPS && prof) tick = lap_clock();
    // The costs were written by other lanes of this workgroup: all its wavefronts share the CU's vector L1
    // (write-through), so a workgroup-scope fence and the barrier make them visible to the replay below.
    __threadfence_block();
    __syncthreads();
    lap(4);  // the first wavefront waiting for the workgroup's last batch
    // gray: the float4 tile {I, gx1, gy1, I} the refinement loops read (stage_tile's second pass), from the plane:
    // the plane lies where the tile goes, so it moves behind it first (where stage_tile stages it).  (Colour: the
    // float4 {B, G, R, 0} tile has been there all along.)
    if constexpr (CH == 1) {
        float *tile = lds + kLutSize, *plane = tile + 4 * tw * th;
        float v[(tw * th + kThreads - 1) / kThreads];
#pragma unroll
        for (int e = 0; e < (tw * th + kThreads - 1) / kThreads; e++) {
            const int k = (int)threadIdx.x + e * kThreads;
            v[e] = k < tw * th ? tile[k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < (tw * th + kThreads - 1) / kThreads; e++) {
            const int k = (int)threadIdx.x + e * kThreads;
            if (k < tw * th) plane[k] = v[e];
        }
        __syncthreads();
        for (int k = threadIdx.x; k < tw * th; k += kThreads) {
            const int ty = k / tw, tx = k - ty * tw;
            float gx1 = 0.0f, gy1 = 0.0f;
            if (tx > 0 && tx < tw - 1 && ty > 0 && ty < th - 1) {
                gx1 = plane[k + 1] - plane[k - 1];
                gy1 = plane[k + tw] - plane[k - tw];
            }
            *reinterpret_cast<float4 *>(tile + 4 * k) = make_float4(plane[k], gx1, gy1, plane[k]);
        }
        __syncthreads();
    }
    lap(5);  // tile for the refinement loops
    sweep_body<BOX, true, true, true, CH>(P, L, lds, norm4, cost, colour, phase, 7u, tune & ~Tune::kPushConsume, true);
    lap(6);  // accept replay, refinement, write-back (the first wavefront's)
    // how long this tile's workgroup lived: the next fused launch of the colour visits the long ones first (tile_order_kernel)
    if (P->tile_clock != nullptr && (threadIdx.x & 63) == 0) {
        typedef __attribute__((address_space(1))) unsigned long long *gp;
        const int gx = (P->cols + kTileW - 1) / kTileW, gy = (P->rows + kSweepTileH - 1) / kSweepTileH;
        const gp q = (gp)P->tile_clock.raw + 2 * ((size_t)colour * (size_t)(gx * gy) + (size_t)((L.y0 / kSweepTileH) * gx + L.x0 / kTileW));
        if (threadIdx.x == 0) q[0] = t_start;
        __hip_atomic_fetch_max(q + 1, lap_clock(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PM_WG_TICKS
    if ((threadIdx.x & 63) == 0 && P->wg_ticks.raw != nullptr) {
        typedef __attribute__((address_space(1))) unsigned long long *gp;
        const gp q = (gp)P->wg_ticks.raw + 4 * (size_t)blockIdx.x;
        const unsigned long long now = wall_clock64();
        if (threadIdx.x == 0) q[0] = wt0;
        __hip_atomic_fetch_max(q + 1, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_min(q + 2, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

// The dispatch order of the next fused launch of `colour` (performance only).  A launch is ~5 workgroups deep per slot, the
// workgroups' durations spread (cv 0.15, the longest twice the mean) and a launch ends with its last workgroup: only 80 %
// of the slots are busy on average (profiles/r06_exp_workgroup_clocks.txt).  A tile's duration correlates 0.8 with its
// duration in the colour's previous fused launch, so the tiles that took more than kTileHeavy times their XCD's mean go
// first, the others keep their place (the spatial order the XCD's L2 lives on).  One workgroup per XCD: workgroup b runs on
// XCD b % 8, and order[b] stays inside that XCD's chunk.  Without durations (the colour's first fused launch of a solve):
// the identity.
constexpr float kTileHeavy = 1.12f;
__global__ __launch_bounds__(kThreads) void tile_order_kernel(const Problem *__restrict__ P, int colour, unsigned tune,
                                                              int *__restrict__ order)
{
    __shared__ float s_sum[4];
    __shared__ int s_cnt[4];
    __shared__ int s_base[2];
    const int gx = (P->cols + kTileW - 1) / kTileW, gy = (P->rows + kSweepTileH - 1) / kSweepTileH;
    const int nblk = gx * gy, xcd = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (nblk - xcd + 7) >> 3;  // workgroups b = 8 i + xcd < nblk
    typedef const __attribute__((address_space(1))) unsigned long long *gp;
    const gp clk = (gp)P->tile_clock.raw + 2 * (size_t)colour * (size_t)nblk;
    auto duration = [&](int i) -> float {
        const TileXY t = tile_of(8 * i + xcd, gx, gy, tune);
        const unsigned long long a = clk[2 * (t.y * gx + t.x)], b = clk[2 * (t.y * gx + t.x) + 1];
        return b > a ? (float)(b - a) : 0.0f;
    };
    float sum = 0.0f;
    for (int i = tid; i < n; i += kThreads) sum += duration(i);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_sum[wave] = sum;
    if (tid < 2) s_base[tid] = 0;
    __syncthreads();
    const float thr = kTileHeavy * (s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / (float)max(n, 1);
    // how many are heavy (thr == 0: no durations, nobody)
    int nh = 0;
    for (int i = tid; i < n; i += kThreads) nh += (thr > 0.0f && duration(i) > thr) ? 1 : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nh += __shfl_xor(nh, d);
    __syncthreads();
    if (lane == 0) s_cnt[wave] = nh;
    __syncthreads();
    const int n_heavy = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    // stable partition, 256 workgroup ids per round: the heavy ones to [0, n_heavy), the others behind them
    for (int i0 = 0; i0 < n; i0 += kThreads) {
        const int i = i0 + tid;
        const bool in = i < n;
        const bool heavy = in && thr > 0.0f && duration(i) > thr;
        const unsigned long long bh = __ballot(heavy), bl = __ballot(in && !heavy);
        __syncthreads();
        if (lane == 0) {
            s_cnt[wave] = (int)__popcll(bh);
            s_sum[wave] = (float)__popcll(bl);
        }
        __syncthreads();
        int hb = s_base[0], lb = s_base[1];
        for (int w = 0; w < wave; w++) {
            hb += s_cnt[w];
            lb += (int)s_sum[w];
        }
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (in) {
            const int pos = heavy ? hb + (int)__popcll(bh & below) : n_heavy + lb + (int)__popcll(bl & below);
            order[8 * pos + xcd] = 8 * i + xcd;
        }
        __syncthreads();
        if (tid == 0) {
            s_base[0] += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_base[1] += (int)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        }
        __syncthreads();
    }
}

}  // namespace pm
