// pm_group.h -- propagation costs of a half-sweep evaluated once per PLANE, not once per (pixel, plane).
//
// What it replaces: the cost evaluations inside gipuma_checkerboard_spatialPropClose_cu / ...Far_cu
// (reference gipuma.cu:1471-1588, 1353-1468; pmCostMultiview_cu :720-806 at :865-872) of the half-sweep that
// follows, for the half-sweeps in which pm::push_kernel (pm_push.h) no longer pays.  The accept tests stay with
// the consumer (sweep_replay in pm_device.h, Tune::kPushConsume), exactly as with pm::push_kernel.
//
// Observation (exact, as in pm_push.h).  The patch cost of plane pi at pixel p in view v is
//     c_v(p, pi) = sum over the window samples q = p + (2i-R, 2j-R) of  w(p, q) * dis_v(q, pi)
// accumulated by fmaf, i outer, j inner (gipuma.cu:633-676), and dis_v(q, pi) does not depend on p.  After the
// first half-sweeps a plane that fits a surface patch has spread over it: the SAME plane, bit for bit, is the
// candidate of many pixels of a tile at once (it is held by several neighbours, each offering it to its up to
// eight consumers), and the windows of those pixels overlap.  On config C the candidates a tile still has to
// evaluate (after the skip rules (A), (D), (H), (S) of sweep_kernel) fall into groups of on average 5 with one
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
// Groups are processed in super-batches -- as many consecutive groups as the sample buffer (38 KB), five strips
// and two tasks per lane allow --; per super-batch the views are looped outside (all its strips for a view, a
// barrier, all its chains, a barrier), so a task's combiner stays in its lane's registers.  The aggregate goes to
// Problem::push_cost[slot][pixel],
// where the half-sweep finds it (Tune::kPushConsume); candidates the skip rules removed get MAXCOST there,
// which the strict < of the accept test (gipuma.cu:868) rejects like their true cost would be.
// Same terms, same order, same roundings as view_cost_pipe + multiview_cost: bit-identical.
//
// Supported: gray window-packed planes with float-encoded offsets, box 11 / 15, best-N with n_best <= 4.
// OPT-IN (GIPUMA_HIP_GROUP_FROM): it halves the propagation stage's instructions and is still slower than the fused
// kernel's propagation rounds (phase clocks).  GIPUMA_HIP_COUNTS=1 reports its phase clocks.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int kGrpMaxTasks = 8 * kThreads;  // 2048
constexpr int kGrpHashSize = 2048;

constexpr int kSbGroups = 128;       // groups per super-batch
constexpr int kSbStripsPerLane = 5;  // strips (sample columns) per lane and view
constexpr int kSbTasksPerLane = 2;   // tasks per lane (their combiners live in registers across the view loop; 4: 256 VGPRs and scratch)

template <int BOX>
struct GroupLayout {  // offsets in 32-bit words into the dynamic LDS array
    static_assert(BOX == 11 || BOX == 15, "instantiated window sizes");
    static constexpr int R = (BOX - 1) / 2, N = R + 1;
    static constexpr int tw = kTileW + 2 * N, th = kSweepTileH + 2 * N;
    static constexpr int max_rows = N + (kSweepTileH - 1) / 2;  // samples per strip: 8 + 7 = 15 for box 15
    static constexpr int max_strips = kSbStripsPerLane * kThreads, max_tasks = kSbTasksPerLane * kThreads;
    // sweep_read_state stages the float4 tile of the sweep kernels at [kLutSize, kLutSize + 4 tw th) and its scalar
    // staging plane -- I alone, clamp-to-edge point samples -- right behind it.  This kernel keeps the PLANE (the
    // gradients are two subtractions per sample, the ones stage_tile does) and puts its tables where the float4
    // tile was.
    static constexpr int meta = kLutSize;                    // [2048 groups][2]: task0 | count << 11 ; rep | bbox << 11
    static constexpr int plane = kLutSize + 4 * tw * th;     // [th][tw] reference texels
    static_assert(meta + 2 * kGrpMaxTasks <= plane, "the group table fits where the float4 tile was");
    // the two task lists: in front of the plane too where there is room (box 15), else behind it
    static constexpr bool lists_in_front = meta + 2 * kGrpMaxTasks + kGrpMaxTasks <= plane;
    static constexpr int btask = lists_in_front ? meta + 2 * kGrpMaxTasks : plane + tw * th;  // [2048] u16: owner | slot << 8
    static constexpr int sorted = btask + kGrpMaxTasks / 2;  // [2048] u16: task indices ordered by group
    static constexpr int misc = (lists_in_front ? plane + tw * th : sorted + kGrpMaxTasks / 2);  // counters
    static constexpr int gtab = misc + 128;                  // [kSbGroups + 1][4]: first strip, first task, first sample of a group
    static constexpr int gplane = gtab + 4 * (kSbGroups + 1);  // [kSbGroups] float4: the planes of the super-batch's groups
    static constexpr int hbuf = gplane + 4 * kSbGroups;      // [kSbGroups][10]: homography + fast-reciprocal flag
    static constexpr int sgroup = hbuf + 10 * kSbGroups;     // [max_strips] u8: group (within the super-batch) of a strip
    static constexpr int tgroup = sgroup + max_strips / 4;   // [max_tasks] u8: ... of a task
    static constexpr int dis = tgroup + max_tasks / 4;       // the sample buffer; while grouping: hash table + group ids
    static constexpr int total = (80 * 1024) / 4;            // two workgroups per CU
    static constexpr int capacity = total - dis;             // samples (of one view) a super-batch may hold
    static_assert(kGrpHashSize + kGrpMaxTasks / 2 <= capacity, "hash table + group ids alias the sample buffer");
    static_assert(capacity >= (N + 15) * max_rows, "the largest possible group fits the sample buffer");
    static_assert(capacity < 65536, "sample offsets fit 16 bits");
};

__device__ __forceinline__ uint32_t plane_hash(float4 pl, int cls)
{
    uint32_t h = __float_as_uint(pl.x) * 0x9E3779B1u;
    h = (h ^ __float_as_uint(pl.y)) * 0x85EBCA77u;
    h = (h ^ __float_as_uint(pl.z)) * 0xC2B2AE3Du;
    h = (h ^ __float_as_uint(pl.w)) * 0x27D4EB2Fu;
    h ^= h >> 15;
    return (h + (uint32_t)cls) & (kGrpHashSize - 1);
}

// dis of the samples (qx, qy0 + 2 r), r = 0 .. nrows - 1, of one view: the per-sample arithmetic of
// view_cost_pipe (getCorrespondingPoint_cu :207-217, the five bilinear taps :251-253, pmCostComputation_shared
// :254-274) without the weight and the accumulation.  `tcol` points at the reference texel of (qx, qy0) in the
// scalar plane (row length tw).
template <bool FAST>
__device__ __forceinline__ void group_strip(const Problem *__restrict__ P, gptr_bytes magic_base,
                                            const float *__restrict__ H, const float *__restrict__ tcol, int tw,
                                            float qx, float qy0, int nrows, float *__restrict__ out)
{
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float pwf = (float)P->pw;
    const float magic_c = kMagicF + (float)(2 * P->pw + 2);
    const float H1 = H[1], H4 = H[4], H7 = H[7];
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
    // (the two requests past the last sample fetch valid, clamped addresses and are dropped)
    WinReq r0 = request(qy0), r1 = request(qy0 + 2.0f);
    float qy = qy0 + 4.0f;
    for (int r = 0; r < nrows; r++, qy += 2.0f) {
        const WinReq cur = r0;
        r0 = r1;
        r1 = request(qy);
        __builtin_amdgcn_sched_barrier(0);
        // {I, gx1, gy1} of the reference texel as stage_tile forms them (gipuma.cu:254-259): central differences
        const float *tq = tcol + 2 * r * tw;
        const float I = tq[0];
        const float gx1 = tq[1] - tq[-1];
        const float gy1 = tq[tw] - tq[-tw];
        const Taps tp5 = taps_u8(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
        const float colDiff = I - tp5.sc;
        const float gradX = gx1 - tp5.gx2;
        const float gradY = gy1 - tp5.gy2;
        const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
        const float colDis = min_abs_nc(colDiff, tau_color);
        out[r] = __builtin_fmaf(alpha, gradDis, oma * colDis);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// grid = tiles of the sweep kernels; `colour`: the colour about to be swept (the consumers); `hist`: rule (H) is
// valid for that half-sweep (the consumer replays only the slots whose producer changed)
template <int BOX>
__global__ __launch_bounds__(kThreads, 2) void group_kernel(const Problem *__restrict__ P,
                                                                      const float4 *__restrict__ norm4,
                                                                      const float *__restrict__ cost, int colour,
                                                                      int hist, unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LY = GroupLayout<BOX>;
    constexpr int R = LY::R, N = LY::N, tw = LY::tw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = P->rows, cols = P->cols, n = P->n_sel;
    const size_t np = (size_t)rows * (size_t)cols;

    // (GIPUMA_HIP_COUNTS: 100 MHz wall-clock ticks per phase, summed over the workgroups, in row 62 of Problem::dbg)
    const bool prof = P->dbg != nullptr && tid == 0;
    unsigned long long tick = prof ? wall_clock64() : 0ull;
    auto lap = [&](int slot) {
        if (prof) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&P->dbg[62 * kDbgSlots + slot], now - tick);
            tick = now;
        }
    };
    // ---- tile, state, skip rules: what sweep_setup does for the half-sweep itself ----
    SweepLane L;
    {
        // rule (H) comes in through `hist`, like in pm::push_kernel; the other rules exactly as the sweep applies them
        const unsigned t2 = (tune & ~(Tune::kPushConsume | Tune::kHistorySkip)) | (hist ? Tune::kHistorySkip : 0u);
        sweep_read_state<BOX, 1>(L, P, lds, norm4, cost, colour, 3u, t2, true);
    }
    lap(0);  // tile + state
    unsigned short *btask = reinterpret_cast<unsigned short *>(lds + LY::btask);
    unsigned short *sorted = reinterpret_cast<unsigned short *>(lds + LY::sorted);
    uint32_t *meta = reinterpret_cast<uint32_t *>(lds + LY::meta);
    int *misc = reinterpret_cast<int *>(lds + LY::misc);
    float *hbuf = lds + LY::hbuf;
    float4 *gplane = reinterpret_cast<float4 *>(lds + LY::gplane);
    float *dbuf = lds + LY::dis;
    uint32_t *hash = reinterpret_cast<uint32_t *>(lds + LY::dis);
    unsigned short *gid_of = reinterpret_cast<unsigned short *>(lds + LY::dis + kGrpHashSize);
    const float *plane = lds + LY::plane;
    const char *lut_magic = (const char *)lds - kMagicBits;

    // every slot the consumer will replay gets a cost: MAXCOST where a skip rule says "cannot be accepted"
    if (L.active) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int nb;
            if (!neighbour(k, L.px, L.py, rows, cols, L.center, nb)) continue;
            const bool replayed = !hist || P->changed[nb] != 0;
            if (replayed && !((L.needmask >> k) & 1u)) P->push_cost[(size_t)k * np + (size_t)L.center] = kMaxCost;
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
        if (tid == 0) misc[8] = 0;  // number of groups
        __syncthreads();  // (the float4 tile under the tables is dead since stage_tile's last barrier)
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

    auto task_plane = [&](int t, int &olx, int &oly, int &nb) -> float4 {
        const unsigned bt = btask[t];
        owner_pixel(L, (int)(bt & 255u), colour, olx, oly);
        const int epx = L.x0 + olx, epy = L.y0 + oly;
        neighbour((int)(bt >> 8), epx, epy, rows, cols, epy * cols + epx, nb);
        return norm4[nb];
    };

    // ---- groups: tasks with bitwise equal planes and the same sample lattice ----
    // pass 1: the first task to claim a hash slot represents its group; gid_of[t] = representative's task index
    for (int t = tid; t < n_tasks; t += kThreads) {
        int olx, oly, nb;
        const float4 pl = task_plane(t, olx, oly, nb);
        const int cls = olx & 1;
        uint32_t h = plane_hash(pl, cls);
        int rep = t;
        for (;;) {
            const uint32_t seen = atomicCAS(&hash[h], 0u, (uint32_t)t + 1u);
            if (seen == 0u) break;  // claimed: this task represents a new group
            const int r = (int)seen - 1;
            int rlx, rly, rnb;
            const float4 rpl = task_plane(r, rlx, rly, rnb);
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
            meta[2 * g] = 0u;  // member count, then fill position
        }
    __syncthreads();
    const int n_groups = misc[8];
    // pass 3: every task learns its group id; member counts
    for (int t = tid; t < n_tasks; t += kThreads) {
        const int g = (int)hash[gid_of[t]];
        gid_of[t] = (unsigned short)g;
        atomicAdd(&meta[2 * g], 1u);
    }
    __syncthreads();
    // exclusive prefix sum of the member counts (8 groups per lane), meta[2g] = task0 | count << 11
    {
        constexpr int per = kGrpMaxTasks / kThreads;
        uint32_t loc[per];
        int sum = 0;
#pragma unroll
        for (int e = 0; e < per; e++) {
            const int g = tid * per + e;
            loc[e] = g < n_groups ? meta[2 * g] : 0u;
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
                meta[2 * g] = (uint32_t)run | (loc[e] << 11);
                meta[2 * g + 1] = 0u;  // fill counter of the scatter below
            }
            run += (int)loc[e];
        }
        __syncthreads();
    }
    // scatter: sorted[task0(g) + k] = t
    for (int t = tid; t < n_tasks; t += kThreads) {
        const int g = gid_of[t];
        const int k = (int)atomicAdd(&meta[2 * g + 1], 1u);
        sorted[(int)(meta[2 * g] & 2047u) + k] = (unsigned short)t;
    }
    __syncthreads();
    // bounding box of a group's pixels, one lane per group: meta[2g+1] = rep task | minlx << 11 | minly << 16 |
    // (ncols - N) << 20 | (nrows - N) << 25
    for (int g = tid; g < n_groups; g += kThreads) {
        const uint32_t m0 = meta[2 * g];
        const int t0 = (int)(m0 & 2047u), cnt = (int)(m0 >> 11);
        int mnx = 255, mxx = 0, mny = 255, mxy = 0;
        for (int k = 0; k < cnt; k++) {
            int olx, oly;
            owner_pixel(L, (int)(btask[sorted[t0 + k]] & 255u), colour, olx, oly);
            mnx = min(mnx, olx);
            mxx = max(mxx, olx);
            mny = min(mny, oly);
            mxy = max(mxy, oly);
        }
        meta[2 * g + 1] = (uint32_t)sorted[t0] | ((uint32_t)mnx << 11) | ((uint32_t)mny << 16) |
                          ((uint32_t)((mxx - mnx) >> 1) << 20) | ((uint32_t)((mxy - mny) >> 1) << 25);
    }
    __syncthreads();

    lap(2);  // grouping, sort, bounding boxes
    // ---- super-batches of consecutive groups: as many as the sample buffer, kSbStripsPerLane strips and
    //      kSbTasksPerLane tasks per lane allow.  Per super-batch the views are looped outside: all its strips for
    //      view v (phase A), then all its chains (phase B) -- two barriers per view, thousands of instructions
    //      between them ----
    uint32_t *gtab = reinterpret_cast<uint32_t *>(lds + LY::gtab);
    unsigned char *sgroup = reinterpret_cast<unsigned char *>(lds + LY::sgroup);
    unsigned char *tgroup = reinterpret_cast<unsigned char *>(lds + LY::tgroup);
    int g_first = 0;
    while (g_first < n_groups) {  // (uniform)
        // cut: groups g_first .. g_first + ng - 1 (lane i of the first two wavefronts looks at group g_first + i)
        {
            const int g = g_first + tid;
            int nc = 0, nt = 0, npnt = 0;
            if (tid < kSbGroups && g < n_groups) {
                const uint32_t m1 = meta[2 * g + 1];
                nc = N + (int)((m1 >> 20) & 31u);
                nt = (int)(meta[2 * g] >> 11);
                npnt = nc * (N + (int)((m1 >> 25) & 15u));
            }
            int ic = nc, it = nt, ip = npnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int uc = __shfl_up(ic, d), ut = __shfl_up(it, d), up = __shfl_up(ip, d);
                if (lane >= d) {
                    ic += uc;
                    it += ut;
                    ip += up;
                }
            }
            if (wave == 0 && lane == 63) {
                misc[16] = ic;
                misc[17] = it;
                misc[18] = ip;
            }
            __syncthreads();
            if (wave == 1) {
                ic += misc[16];
                it += misc[17];
                ip += misc[18];
            }
            const bool fits = nc > 0 && ic <= LY::max_strips && it <= LY::max_tasks && ip <= LY::capacity;
            const unsigned long long fb = __ballot(fits);
            // groups are taken while they fit: the run of set bits from bit 0 of the first wavefront on
            if (wave < 2 && lane == 0) misc[20 + wave] = fb == ~0ull ? 64 : (int)__builtin_ctzll(~fb);
            if (tid < kSbGroups) {  // exclusive prefixes; the entry behind the last group taken = the totals
                gtab[4 * (tid + 1) + 0] = (uint32_t)ic;
                gtab[4 * (tid + 1) + 1] = (uint32_t)it;
                gtab[4 * (tid + 1) + 2] = (uint32_t)ip;
            }
            if (tid == 0) gtab[0] = gtab[1] = gtab[2] = 0u;
            __syncthreads();
        }
        // (a single group always fits: at most 23 strips of 15 samples and the 128 pixels of its lattice)
        const int ng = misc[20] < 64 ? misc[20] : 64 + misc[21];
        const int n_strips = (int)gtab[4 * ng + 0], n_btasks = (int)gtab[4 * ng + 1];
        // tables: which group a strip / a task of the super-batch belongs to; the groups' planes
        if (tid < ng) {
            const uint32_t m1 = meta[2 * (g_first + tid) + 1];
            const int s0 = (int)gtab[4 * tid + 0], s1 = (int)gtab[4 * tid + 4];
            const int t0 = (int)gtab[4 * tid + 1], t1 = (int)gtab[4 * tid + 5];
            for (int k = s0; k < s1; k++) sgroup[k] = (unsigned char)tid;
            for (int k = t0; k < t1; k++) tgroup[k] = (unsigned char)tid;
            int rlx, rly, rnb;
            gplane[tid] = task_plane((int)(m1 & 2047u), rlx, rly, rnb);
        }
        __syncthreads();
        // this lane's tasks (kept in registers across the views)
        int t_off[kSbTasksPerLane], t_slot[kSbTasksPerLane], t_center[kSbTasksPerLane];  // sample-buffer offset, slot, pixel
        int t_nrows[kSbTasksPerLane];
        const float *t_tp0[kSbTasksPerLane];
        ViewCombiner<true> comb[kSbTasksPerLane];
#pragma unroll
        for (int k = 0; k < kSbTasksPerLane; k++) {
            const int t = tid + k * kThreads;
            t_off[k] = 0;
            t_slot[k] = 0;
            t_center[k] = 0;
            t_nrows[k] = N;
            t_tp0[k] = plane + (L.hh * tw + L.hw);
            if (t < n_btasks) {
                const int j = tgroup[t];
                const uint32_t m0 = meta[2 * (g_first + j)], m1 = meta[2 * (g_first + j) + 1];
                const unsigned bt = btask[sorted[(int)(m0 & 2047u) + (t - (int)gtab[4 * j + 1])]];
                int olx, oly;
                owner_pixel(L, (int)(bt & 255u), colour, olx, oly);
                const int nrows = N + (int)((m1 >> 25) & 15u);
                t_nrows[k] = nrows;
                t_off[k] = (int)gtab[4 * j + 2] + ((olx - (int)((m1 >> 11) & 31u)) >> 1) * nrows + ((oly - (int)((m1 >> 16) & 15u)) >> 1);
                t_slot[k] = (int)(bt >> 8);
                t_center[k] = (L.y0 + oly) * cols + (L.x0 + olx);
                t_tp0[k] = plane + ((oly + L.hh) * tw + (olx + L.hw));
            }
        }
        // homography of (group plane, view) -> hbuf; [9] = the fast reciprocal is exact on the whole box
        auto group_h = [&](int j, int v) {
            const uint32_t m1 = meta[2 * (g_first + j) + 1];
            float H[9];
            homography(P->rc.K_inv, P->view[v], gplane[j], H);
            const int bx0 = L.x0 + (int)((m1 >> 11) & 31u) - R, by0 = L.y0 + (int)((m1 >> 16) & 15u) - R;
            const int bx1 = bx0 + 2 * (N - 1 + (int)((m1 >> 20) & 31u)), by1 = by0 + 2 * (N - 1 + (int)((m1 >> 25) & 15u));
            const bool safe = window_z_safe(H, (float)bx0, (float)bx1, (float)by0, (float)by1);
            float *h = hbuf + 10 * j;
#pragma unroll
            for (int k = 0; k < 9; k++) h[k] = H[k];
            h[9] = safe ? 1.0f : 0.0f;
        };
        if (tid < ng) group_h(tid, 0);
        __syncthreads();
        lap(3);  // super-batch cut, tables, task geometry, first homographies
        for (int v = 0; v < n; v++) {
            const ViewCam &vc = P->view[v];
            const gptr_bytes base = (gptr_bytes)((uintptr_t)vc.packed - (uintptr_t)kMagicBits);
            // phase A: dis of every sample column of the super-batch
#ifndef PM_GROUP_EXP_STRIP_REPEAT
#define PM_GROUP_EXP_STRIP_REPEAT 1  // (timing experiments: phase A run this many times, same results)
#endif
            for (int rep = 0; rep < PM_GROUP_EXP_STRIP_REPEAT; rep++)
            for (int s0 = 0; s0 < n_strips; s0 += kThreads) {  // (uniform trip count)
                const int sidx = s0 + tid;
                const bool has = sidx < n_strips;
                const int j = has ? (int)sgroup[sidx] : 0;
                const uint32_t m1 = meta[2 * (g_first + j) + 1];
                const int nrows = N + (int)((m1 >> 25) & 15u);
                const int c = has ? sidx - (int)gtab[4 * j + 0] : 0;
                const int s_lx = (int)((m1 >> 11) & 31u) - R + 2 * c, s_ly = (int)((m1 >> 16) & 15u) - R;
                const float *h = hbuf + 10 * j;
                float H[9];
#pragma unroll
                for (int k = 0; k < 9; k++) H[k] = h[k];
                const bool safe = h[9] != 0.0f || !has;
                const float *tcol = plane + ((s_ly + L.hh) * tw + (s_lx + L.hw));
                float *out = dbuf + (int)gtab[4 * j + 2] + c * nrows;
                if (__all(safe))
                    group_strip<true>(P, base, H, tcol, tw, (float)(L.x0 + s_lx), (float)(L.y0 + s_ly), has ? nrows : 0, out);
                else
                    group_strip<false>(P, base, H, tcol, tw, (float)(L.x0 + s_lx), (float)(L.y0 + s_ly), has ? nrows : 0, out);
            }
            __syncthreads();
            lap(4);  // phase A
            // phase B: the reference's summation order over each task's own window ...
#pragma unroll
            for (int k = 0; k < kSbTasksPerLane; k++) {
                if (tid + k * kThreads < n_btasks) {
                    const float centre = t_tp0[k][0];
                    float cst = 0.0f;
#ifndef PM_GROUP_EXP_CHAIN_REPEAT
#define PM_GROUP_EXP_CHAIN_REPEAT 1  // (timing experiments: the chain run this many times, same results)
#endif
                    for (int rep = 0; rep < PM_GROUP_EXP_CHAIN_REPEAT; rep++) {
                    cst = 0.0f;
                    const float *dcol = dbuf + t_off[k];
                    const float *tcol = t_tp0[k] + (-R * tw - R);
                    const int nrows = t_nrows[k];
                    __builtin_amdgcn_sched_barrier(0);
                    for (int i = 0; i < N; i++, dcol += nrows, tcol += 2) {
#pragma unroll
                        for (int jj = 0; jj < N; jj++) {
                            const float colorDis = __builtin_fabsf(tcol[2 * jj * tw] - centre);
                            const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
                            cst = __builtin_fmaf(w, dcol[jj], cst);
                        }
                    }
                    }
                    comb[k].add(cst, v, nullptr);
                }
            }
            // ... and the next view's homographies (read by phase A only)
            if (v + 1 < n && tid < ng) group_h(tid, v + 1);
            __syncthreads();
            lap(5);  // phase B + next homographies
        }
#pragma unroll
        for (int k = 0; k < kSbTasksPerLane; k++)
            if (tid + k * kThreads < n_btasks)
                P->push_cost[(size_t)t_slot[k] * np + (size_t)t_center[k]] = comb[k].finish(P, n, nullptr);
        g_first += ng;
    }
}

}  // namespace pm
