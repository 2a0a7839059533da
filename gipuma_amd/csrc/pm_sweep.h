// pm_sweep.h -- the kernels of a solve: tile order and staging, random initialisation (gipuma_init_cu2), the fused
// half-sweep (close + far propagation + refinement of one colour: gipuma.cu:1353-1711) with its exact skip rules,
// task lists, accept replay, the two-phase refinement by (candidate, view) items, the column-per-lane variant,
// and the final conversion (gipuma_compute_disp).  Part of the device code of the PatchMatch path (pm_device.h).
#pragma once
#include "pm_core.h"
#include "pm_cost.h"
#include "pm_prefilter.h"

namespace pm {

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
// (Measured and refuted in round 6, profiles/r06_exp_xcd_interleave.txt: an XCD's tiles as several thinner bands spread over
// the frame instead of one chunk -- 76.3 -> 79.6 ms per view with bands of 3 tile rows, level only at the default height.)
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
        // The frame's border tiles take ~30 % longer than the others (profiles/r06_exp_workgroup_clocks.txt), and in the plain
        // column order every XCD's chunk ENDS with its band's last columns -- the launch then waits for a few long
        // workgroups: the last tile column is visited right after the first one (Tune::kPlainColumnOrder: in place).
        const int cx = rem / h;
        o.x = (tune & Tune::kPlainColumnOrder) || gx < 3 ? cx : cx == 0 ? 0 : cx == 1 ? gx - 1 : cx - 1;
        o.y = band * bh + rem % h;
    }
    return o;
}

// stage the reference tile (+halo) and the weight table; the tile holds clamp-to-edge point
// samples exactly like the reference's (gipuma.cu:1393-1402, 1513-1522)
// (PAD: extra texels per tile row, so that lanes two tile rows apart do not share LDS banks)
// (PLANE_ONLY, gray: only the scalar plane of point samples, AT the float4 tile's place -- for kernels that form
//  the gradients per sample and want the LDS, pm_group.h)
template <int BOX, int CH, int PAD = 0, bool PLANE_ONLY = false>
__device__ __forceinline__ void stage_tile(const Problem *__restrict__ P, float *lds, int x0, int y0,
                                           int tile_h, const Win<BOX> &win, bool want_lut)
{
    static_assert(!PLANE_ONLY || (CH == 1 && PAD == 0), "plane-only staging: gray tiles");
    const int hw = win.halo_w(), hh = win.halo_h();
    const int tw = kTileW + 2 * hw, th = tile_h + 2 * hh;
    const int tws = tw + PAD;  // row stride of the float4 tile
    const gptr_f32 ref = (gptr_f32)P->ref.raw;
    float *tile = lds + lut_size<CH>();  // float4 per texel
    // gray: the scalar image goes to a scratch plane behind the float4 tile first, so that the
    // central differences can be formed once per tile instead of once per sample
    float *plane = PLANE_ONLY ? tile : tile + 4 * tws * th;
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
    if (CH == 1 && !PLANE_ONLY) {
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
        norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)] = pl;
    } else {
        pl = norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)];
    }
    float c;
    if (tune & Tune::kNoInterior)
        c = multiview_cost<BOX, U8, false, COMBINE_REG, CH>(P, tp0, tw, lds, cv, px, py, pl, win);
    else
        c = multiview_cost<BOX, U8, true, COMBINE_REG, CH>(P, tp0, tw, lds, cv, px, py, pl, win);
    cost[PM_AT(P, center, PM_NP(P), kChkCost)] = c;
}

// The 256 (pixel, plane) pairs of a column-per-lane step ordered by DISPARITY bucket (performance only: a pair's cost does
// not depend on when it is evaluated): the 32 pairs a workgroup evaluates together then see the source views at similar
// offsets along the epipolar lines, so their windows fall into a region the CU's vector L1 holds (see pm_push.h).
// `order[k]` = lane whose pair is evaluated k-th; `counters`: kDispBuckets ints.  All lanes call it (two barriers).
#ifndef PM_DISP_BUCKETS
#define PM_DISP_BUCKETS 16
#endif
constexpr int kDispBuckets = PM_DISP_BUCKETS;  // (<= 32: the counters of the sweep kernels are the 32 ints of SweepLane::wcnt)
static_assert(kDispBuckets <= 32, "bucket counters");
__device__ __forceinline__ void disparity_order(const Problem *__restrict__ P, float depth, bool sort, unsigned short *order,
                                                int *counters)
{
    const int tid = threadIdx.x;
    if (!sort) {
        order[tid] = (unsigned short)tid;
        __syncthreads();
        return;
    }
    if (tid < kDispBuckets) counters[tid] = 0;
    const float disp = disp_depth(P->rc.f, P->rc.baseline, depth);
    const float t = (disp - P->min_disp) / (P->max_disp - P->min_disp) * (float)kDispBuckets;
    const int bucket = (int)__builtin_fminf(__builtin_fmaxf(t, 0.0f), (float)(kDispBuckets - 1));  // (NaN -> 0)
    __syncthreads();
    const int rank = atomicAdd(&counters[bucket], 1);
    __syncthreads();
    int start = 0;
#pragma unroll
    for (int b = 0; b < kDispBuckets; b++) start += b < bucket ? counters[b] : 0;
    order[start + rank] = (unsigned short)tid;
    __syncthreads();
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
            norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)] = pl;
        } else {
            pl = norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)];
        }
    }
    candbuf[threadIdx.x] = pl;
    // (evaluation order: by disparity bucket, see disparity_order; the order array and its counters lie behind the results)
    unsigned short *order = reinterpret_cast<unsigned short *>(bres + kThreads);
    disparity_order(P, depth_from_plane(P->rc, pl, min(px, P->cols - 1), min(py, P->rows - 1)), !(tune & Tune::kNoDispSort), order,
                    reinterpret_cast<int *>(bres + kThreads + kThreads / 2));
    constexpr int kColGroup = col_group<BOX>(), kColTasks = col_tasks<BOX>();
    const int grp = threadIdx.x / kColGroup, col = threadIdx.x % kColGroup;
    for (int r = 0; r < kThreads / kColTasks; r++) {
        const int owner = (int)order[r * kColTasks + grp];
        // pixels outside the image evaluate their dummy plane at the clamped position (never stored)
        const int epx = min(x0 + (owner & 31), P->cols - 1), epy = min(y0 + (owner >> 5), P->rows - 1);
        const float4 ecand = candbuf[owner];
        const float *etp0 = tile + (((epy - y0) + hh) * tw + ((epx - x0) + hw)) * 4;
        const float c = multiview_cost_cols<BOX, false, CH>(P, etp0, tw, lds, cv, epx, epy, ecand, col);
        if (col == 0) bres[owner] = c;
    }
    __syncthreads();
    if (active) cost[PM_AT(P, center, PM_NP(P), kChkCost)] = bres[threadIdx.x];
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
//     (S) (where the host gives the pixels a ring, Problem::seen_ring: colour sessions) a plane this pixel's
//         propagation evaluated before -- one of its last kSeenRing evaluated candidates: its cost F is a pure
//         function of (pixel, plane); it was then rejected against a cost that has only decreased since (or for its
//         depth, which is a pure function too), or accepted -- and the pixel's cost has been <= F ever since.  This
//         does not need the state invariant, only that the pixel's cost never increases between the two
//         half-sweeps: the host clears the rings whenever planes are (re-)installed.  A plane that spreads over a
//         patch reaches a pixel that turned it down again and again, through every neighbour that adopts it
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
};

// tile staging, lane -> pixel mapping, state read (gipuma.cu:1527-1530) and the exact skipping rules:
// leaves L.needmask = the candidate slots of this lane's pixel that must be evaluated
template <int BOX, int CH, int PAD = 0, bool PLANE_ONLY = false>
__device__ __forceinline__ void sweep_read_state(SweepLane &L, const Problem *__restrict__ P, float *lds,
                                                 const float4 *__restrict__ norm4, const float *__restrict__ cost,
                                                 int colour, unsigned stages, unsigned tune, bool want_lut,
                                                 const int *__restrict__ order = nullptr)
{
    const Win<BOX> win(P);
    const RefCam &rc = P->rc;
    const int rows = P->rows, cols = P->cols;
    const int gx = (cols + kTileW - 1) / kTileW;
    const int gy = (rows + kSweepTileH - 1) / kSweepTileH;
    // (`order`: the fused launches' dispatch order, Problem::tile_order -- workgroup b does the tile of workgroup order[b])
    const int wg = order != nullptr ? ((const __attribute__((address_space(1))) int *)order)[blockIdx.x] : (int)blockIdx.x;
    const TileXY txy = tile_of(wg, gx, gy, tune);
    L.x0 = txy.x * kTileW;
    L.y0 = txy.y * kSweepTileH;
    stage_tile<BOX, CH, PAD, PLANE_ONLY>(P, lds, L.x0, L.y0, kSweepTileH, win, want_lut);
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
    const bool history = (tune & Tune::kHistorySkip) != 0;
    if (L.active) {
        const float4 pl = norm4[PM_AT(P, L.center, PM_NP(P), kChkNorm4)];
        L.pl = pl;
        L.cst = cost[PM_AT(P, L.center, PM_NP(P), kChkCost)];
        L.depth = depth_from_plane(rc, pl, L.px, L.py);
        float4 cands[8];
        unsigned valid = 0, needmask = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int nb;
            const bool ok = neighbour(k, L.px, L.py, rows, cols, L.center, nb) && (stages & (k < 4 ? 1u : 2u));
            if (ok) {
                cands[k] = norm4[PM_AT(P, nb, PM_NP(P), kChkNorm4)];
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
                    if (!history || P->changed[PM_AT(P, nb, PM_NP(P), kChkFlags)] != 0) needmask |= 1u << k;
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
                    if (P->changed[PM_AT(P, nb, PM_NP(P), kChkFlags)] == 0) fresh = false;
                }
#pragma unroll
                for (int j = 0; j < k; j++)
                    if (fresh && ((valid >> j) & 1u) && same_bits(cands[k], cands[j])) fresh = false;  // (D)
                if (fresh) needmask |= 1u << k;
            }
            if (P->seen_ring != nullptr && !(tune & Tune::kNoSeen)) {  // (S)
                const size_t np = (size_t)rows * (size_t)cols;
                const unsigned st = P->seen_pos[PM_AT(P, L.center, PM_NP(P), kChkFlags)];
                const int cnt = (st & 8u) ? kSeenRing : (int)(st & 7u);
#pragma unroll
                for (int a = 0; a < kSeenRing; a++) {
                    if (a < cnt && needmask != 0u) {
                        const float4 e = P->seen_ring[PM_AT(P, (size_t)a * np + (size_t)L.center, (size_t)kSeenRing * np, kChkFlags)];
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (((needmask >> k) & 1u) && same_bits(cands[k], e)) needmask &= ~(1u << k);
                    }
                }
                unsigned pos = st & 7u, full = st & 8u;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if ((needmask >> k) & 1u) {
                        P->seen_ring[PM_AT(P, (size_t)pos * np + (size_t)L.center, (size_t)kSeenRing * np, kChkFlags)] = cands[k];
                        pos = (pos + 1u) & 7u;
                        if (pos == 0u) full = 8u;
                    }
                }
                P->seen_pos[PM_AT(P, L.center, PM_NP(P), kChkFlags)] = (unsigned char)(pos | full);
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
            const float4 cand = norm4[PM_AT(P, nb, PM_NP(P), kChkNorm4)];
            const float c = pushed ? P->push_cost[PM_AT(P, (size_t)k * np + (size_t)L.center, 8 * np, kChkPushCost)] : L.bres[k * kThreads + threadIdx.x];
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
        const gptr_bytes base = (gptr_bytes)((uintptr_t)vc.packed.raw - (uintptr_t)kMagicBits);
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
                                                  const uint32_t (&ord)[kLbRegDwords], int *items_short = nullptr)
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
                    const gptr_bytes base = (gptr_bytes)((uintptr_t)P->view[v].packed.raw - (uintptr_t)kMagicBits);
                    if (__all(safe))
                        lb = lb_item<BOX, true>(P, base, H, tp0, L.tw, lut, L.px, L.py, ordp, np, ord, lbk >> 1, &lbs);
                    else
                        lb = lb_item<BOX, false>(P, base, H, tp0, L.tw, lut, L.px, L.py, ordp, np, ord, lbk >> 1, &lbs);
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
        constexpr int kLbFirst = lb_max<BOX>();  // no measurement yet: the whole list
        if (P->lb_k > 0) {
            lbk = min(P->lb_k & ~1, lb_max<BOX>());
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
                lbk = max(4, min(lbk, lb_max<BOX>()));
            }
        }
        ordp += L.active ? (size_t)L.center : 0;
    }
    uint32_t ord[kLbRegDwords] = {};
    if constexpr (CH == 1 && lb_in_registers<BOX>()) {
        if (lbk > 0) {
            const size_t npx = (size_t)P->rows * (size_t)P->cols;
#pragma unroll
            for (int d = 0; d < lb_max<BOX>() / 2; d++) ord[d] = ordp[(size_t)d * npx];
        }
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
                                                   lbk, ordp, ord, probe && pass == 0 ? &items_short : nullptr);
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
        dbg_add(&d[kDbgItemsOpen], (unsigned long long)items_left);
        dbg_add(&d[kDbgRedo], (unsigned long long)n_redo);
    }
    if (P->dbg != nullptr) {
        const unsigned n_cand = (unsigned)__popcll(__ballot(do_eval));
        if ((threadIdx.x & 63u) == 0u) {
            unsigned long long *d = P->dbg + (size_t)(phase & 63u) * kDbgSlots;
            dbg_add(&d[kDbgCands], (unsigned long long)n_cand);
            dbg_add(&d[kDbgItems], (unsigned long long)n_cand * (unsigned)P->n_sel);
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
#ifndef PM_SWEEP_WG_C4
#define PM_SWEEP_WG_C4 4  // ... and the colour one (4: 128 VGPRs)
#endif
// Everything of a fused half-sweep after its set-up (sweep_setup, or pm::sweep_group_kernel's own): the propagation
// rounds over the workgroup's task list, the accept replay, the refinement steps, the write-back.  `pushed`: the
// costs of the candidates in L.needmask are in Problem::push_cost (there is no task list).
template <int BOX, bool U8, bool COMBINE_REG, bool INTERIOR, int CH>
__device__ __forceinline__ void sweep_body(const Problem *__restrict__ P, SweepLane &L, float *lds,
                                           float4 *__restrict__ norm4, float *__restrict__ cost, int colour,
                                           uint32_t phase, unsigned stages, unsigned tune, bool pushed)
{
    const Win<BOX> win(P);
    const int rows = P->rows, cols = P->cols;
    const int prop_rounds = (L.n_tasks + kThreads - 1) / kThreads;
    if (P->dbg != nullptr) {
        unsigned long long *d = P->dbg + (size_t)(phase & 63u) * kDbgSlots;
        if (threadIdx.x == 0) dbg_add(&d[kDbgTasks], (unsigned long long)L.n_tasks);
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
            sweep_replay(L, P, norm4, pushed);
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
                cand = norm4[PM_AT(P, nb, PM_NP(P), kChkNorm4)];
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
        cost[PM_AT(P, L.center, PM_NP(P), kChkCost)] = L.cst;
        norm4[PM_AT(P, L.center, PM_NP(P), kChkNorm4)] = L.pl;
        P->changed[PM_AT(P, L.center, PM_NP(P), kChkFlags)] = (unsigned char)(L.chg | ((tune & Tune::kAccumChanged) ? P->changed[PM_AT(P, L.center, PM_NP(P), kChkFlags)] : 0u));
    }
}

template <int BOX, bool U8, bool COMBINE_REG, bool INTERIOR, int CH>
__global__ __launch_bounds__(kThreads, U8 ? (CH == 4 ? PM_SWEEP_WG_C4 : PM_SWEEP_WG) : 1) void sweep_kernel(const Problem *__restrict__ P,
                                                         float4 *__restrict__ norm4, float *__restrict__ cost,
                                                         int colour, uint32_t phase, unsigned stages,
                                                         unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    SweepLane L;
    sweep_setup<BOX, CH>(L, P, lds, norm4, cost, colour, stages, tune, U8);
    sweep_body<BOX, U8, COMBINE_REG, INTERIOR, CH>(P, L, lds, norm4, cost, colour, phase, stages, tune,
                                                   (tune & Tune::kPushConsume) != 0);
}

// The same half-sweep with the column-per-lane evaluation (see view_cost_cols): state, candidate
// selection, task list, accept replay and refinement candidates are computed per pixel by its owner
// lane exactly as in sweep_kernel (the shared helpers above); only the cost evaluations are done by
// groups of col_group<BOX>() lanes, col_tasks<BOX>() (pixel, plane) pairs at a time, exchanging planes and costs
// through LDS.  Gray packed planes with float-encoded offsets and a compile-time box only (the host
// uses it for box 15, whose 8 columns fill a group of 8, and for box 25: 13 of 16 lanes).
#ifndef PM_COLS_WG
#define PM_COLS_WG 1  // workgroups per CU the column-per-lane sweep kernel is compiled for (1: the compiler's choice of registers)
#endif
template <int BOX, bool COMBINE_REG, int CH = 1>
__global__ __launch_bounds__(kThreads, PM_COLS_WG) void sweep_cols_kernel(const Problem *__restrict__ P,
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
        const float4 cand = norm4[PM_AT(P, nb, PM_NP(P), kChkNorm4)];
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
        // (evaluation order: by disparity bucket, see disparity_order; rows 1.. of bres are free during the refinement)
        unsigned short *order = reinterpret_cast<unsigned short *>(L.bres + kThreads);
        disparity_order(P, L.active ? d_new : P->rc.depth_max, !(tune & Tune::kNoDispSort), order, L.wcnt);
        for (int r = 0; r < kThreads / kColTasks; r++) {
            const int owner = (int)order[r * kColTasks + grp];
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
        cost[PM_AT(P, L.center, PM_NP(P), kChkCost)] = L.cst;
        norm4[PM_AT(P, L.center, PM_NP(P), kChkNorm4)] = L.pl;
        P->changed[PM_AT(P, L.center, PM_NP(P), kChkFlags)] = (unsigned char)(L.chg | ((tune & Tune::kAccumChanged) ? P->changed[PM_AT(P, L.center, PM_NP(P), kChkFlags)] : 0u));
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
    const float4 pl = norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)];
    Vec3 v = {pl.x, pl.y, pl.z};
    const Vec3 w = matvec(P->rc.R_orig_inv, v);
    float depth = 0.0f;
    if (cost[PM_AT(P, center, PM_NP(P), kChkCost)] != kMaxCost) depth = depth_from_plane(P->rc, pl, px, py);
    norm4[PM_AT(P, center, PM_NP(P), kChkNorm4)] = make_float4(w.x, w.y, w.z, depth);
}

}  // namespace pm
