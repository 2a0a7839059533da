// pm_refine_rows.h -- one checkerboard half-sweep with ROW-PER-LANE refinement evaluations.
//
// Why.  The candidates of planeRefinement_cu (reference gipuma.cu:928-994) are random perturbations
// of each pixel's own plane: with one lane per pixel the 64 lanes of a window load sit in 64
// unrelated cache lines, and the refinement rounds of pm::sweep_kernel run at ~55 % of their VALU
// bound behind the vector L1's line fills (profiles/).  Its early termination
// (pm::multiview_cost) also only pays when all 64 pixels of a wavefront agree.
//
// Here 8 consecutive lanes evaluate ONE (pixel, plane) pair, lane r taking window ROW r, and walk the
// window columns together -- the reference's summation order is columns outer, rows inner
// (gipuma.cu:633-676), so after every column the group holds an exact prefix of the view cost and can
// stop: early termination per group of 8 lanes (8 pixels per wavefront instead of 64).  The 8 lanes
// of a group sample one source column 2 rows apart; source views are read from a second,
// COLUMN-MAJOR window-packed copy (pack_t_kernel) in which those 8 windows are 8 bytes apart, i.e.
// in one cache line: a wavefront touches 8-16 lines per load instead of 64.
//
// The row order inside a column is kept by a relay: in stage s every lane computes
// fmaf(w, dis, value of the lane above) -- DPP row_shr:1, lane 0 taking the previous column's total
// from lane N-1 (row_shl:N-1) -- so that after stage s lane s holds the exact prefix.  Same terms,
// same order, same roundings as view_cost_pipe: bit-identical.
//
// Propagation rounds are those of pm::sweep_kernel (one lane per pixel, compacted task list).
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int kRowGroup = 8;                      // lanes per (pixel, plane) pair
constexpr int kRowTasks = kThreads / kRowGroup;   // pairs evaluated concurrently by a workgroup
constexpr int kRowsTilePad = 1;                   // tile row stride 49 texels: lanes 2 rows apart hit different banks

// column-major window-packed plane:  VT[X][Y] (one 32-bit word) = bytes { Pd(Y, X + c) : c = 0..3 },
// X in [0, cols+3), Y in [0, ph = rows+8), Pd = the image with a 3-texel replicated border as in
// pack_kernel.  The 4x4 window whose top-left texel is Pd(Y, X) is VT[X][Y..Y+3]: 16 contiguous bytes
// (word r = row Y+r, byte c = column X+c).  grid = (ceil(ph/256), cols+3)
__global__ __launch_bounds__(kThreads) void pack_t_kernel(const float *__restrict__ img, int rows, int cols,
                                                          int pitch, int ph, uint32_t *__restrict__ packed_t)
{
    const int Y = blockIdx.x * kThreads + threadIdx.x;
    const int X = blockIdx.y;
    if (Y >= ph) return;
    const int y = clampi(Y - 3, 0, rows - 1);
    uint32_t w = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int x = clampi(X + c - 3, 0, cols - 1);
        w |= ((uint32_t)img[y * pitch + x] & 0xffu) << (8 * c);
    }
    packed_t[(size_t)X * ph + Y] = w;
}

// the five bilinear taps from a column-major window: word r = row Y+r, byte c = column X+c
__device__ __forceinline__ Taps taps_u8_t(float a, float b, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    return taps12(a, b, ub1(w0), ub2(w0), ub0(w1), ub1(w1), ub2(w1), ub3(w1), ub0(w2), ub1(w2), ub2(w2),
                  ub3(w2), ub1(w3), ub2(w3));
}

// pmCost_shared + pmCostComputation_shared (gipuma.cu:585-680, 223-277) of one source view for the
// group's (pixel, plane) pair; the result is exact in lane N-1 of the group.  With ET the group
// stops after the first column at which its prefix has reached `tau` -- the whole wavefront leaves
// the view once all of its groups have.
template <int BOX, bool FAST, bool ET>
__device__ __forceinline__ float view_cost_rows(const Problem *__restrict__ P, const ViewCam &vc,
                                                const float *__restrict__ H, const float *__restrict__ tp0,
                                                int tws, const float *__restrict__ lut, int px, int py, int row,
                                                float tau)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(BOX > 0 && N <= kRowGroup, "one lane per window row");
    const float colsf = (float)P->cols, rowsf = (float)P->rows;
    const float alpha = P->alpha, oma = 1.f - P->alpha;
    const float tau_color = P->tau_color, tau_gradient = P->tau_gradient;
    const float centre = tp0[0];
    const float phf = (float)P->ph;
    const float magic_c = kMagicF + (float)(2 * P->ph + 2);
    const gptr_bytes magic_base = (gptr_bytes)((uintptr_t)vc.packed_t - (uintptr_t)kMagicBits);
    const char *lut_magic = (const char *)lut - kMagicBits;
    const int myrow = row < N ? row : N - 1;  // spare lanes of a smaller box shadow the last row
    const float qy = (float)(py - R + 2 * myrow);
    const float H1qy = H[1], H4qy = H[4], H7qy = H[7];
    const float qx0 = (float)(px - R);

    auto request = [&](int c) -> WinReq {
        // getCorrespondingPoint_cu, gipuma.cu:207-217, same fmaf nesting as view_cost_pipe
        const float qx = qx0 + (float)(2 * c);
        const float X = __builtin_fmaf(H1qy, qy, __builtin_fmaf(H[0], qx, H[2]));
        const float Y = __builtin_fmaf(H4qy, qy, __builtin_fmaf(H[3], qx, H[5]));
        const float Z = __builtin_fmaf(H7qy, qy, __builtin_fmaf(H[6], qx, H[8]));
        const float rz = recip<FAST>(Z);
        const float sx = X * rz, sy = Y * rz;
        const float fx0 = __builtin_floorf(sx), fy0 = __builtin_floorf(sy);
        WinReq r;
        r.a = sx - fx0;
        r.b = sy - fy0;
        const float Xc = __builtin_amdgcn_fmed3f(fx0, -2.0f, colsf);
        const float Yc = __builtin_amdgcn_fmed3f(fy0, -2.0f, rowsf);
        const uint32_t off = __float_as_uint(__builtin_fmaf(Xc, phf, Yc + magic_c));
        r.w = *(gptr_u32x4)(magic_base + off);
        return r;
    };

    const float *trow = tp0 + 4 * ((2 * myrow - R) * tws - R);  // texel (-R, row) of the window
    constexpr int PD = 3 < N ? 3 : N;  // window requests in flight
    WinReq req[PD];
#pragma unroll
    for (int p = 0; p < PD; p++) req[p] = request(p);
    float acc = 0.0f;
    // lanes N-1 of the 8 groups of a wavefront
    constexpr unsigned long long kLastRows = 0x0101010101010101ull << (N - 1);
#pragma unroll
    for (int c = 0; c < N; c++) {
        const WinReq cur = req[c % PD];
        if (c + PD < N) req[c % PD] = request(c + PD);
        __builtin_amdgcn_sched_barrier(0);
        const float4 t4 = *reinterpret_cast<const float4 *>(trow + 8 * c);
        // weight_cu, gipuma.cu:186-193
        const float colorDis = __builtin_fabsf(t4.x - centre);
        const float w = *(const float *)(lut_magic + __float_as_uint(colorDis + kMagicF));
        const Taps tp5 = taps_u8_t(cur.a, cur.b, cur.w.x, cur.w.y, cur.w.z, cur.w.w);
        // pmCostComputation_shared, gipuma.cu:251-274
        const float colDiff = t4.w - tp5.sc;
        const float gradX = t4.y - tp5.gx2;
        const float gradY = t4.z - tp5.gy2;
        const float gradDis = min_nc((__builtin_fabsf(gradX) + __builtin_fabsf(gradY)) * 0.0625f, tau_gradient);
        const float colDis = min_abs_nc(colDiff, tau_color);
        const float dis = __builtin_fmaf(alpha, gradDis, oma * colDis);
        // relay down the rows of this column (reference order: rows inner)
#pragma unroll
        for (int s = 0; s < N; s++) {
            float in;
            if (s == 0)  // lane 0 continues from the previous column's total in lane N-1
                in = c == 0 ? 0.0f
                            : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x100 + (N - 1), 0xf,
                                                                         0xf, false));
            else
                in = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x111, 0xf, 0xf, false));
            acc = __builtin_fmaf(w, dis, in);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ET && c + 1 < N) {
            const unsigned long long live = __ballot(true) & kLastRows;  // (a redo pass runs some groups only)
            if ((__ballot(acc >= tau) & live) == live) break;
        }
    }
    return acc;  // exact (or an early-exit lower bound >= tau) in lane N-1 of the group
}

// pmCostMultiview_cu for the group's pair, early termination as in multiview_cost; exact in every lane
template <int BOX, bool COMBINE_REG, bool ET>
__device__ __forceinline__ float multiview_cost_rows(const Problem *__restrict__ P, const float *__restrict__ tp0,
                                                     int tws, const float *__restrict__ lut, float *cv, int px,
                                                     int py, float4 pl, int row, bool et_on, float thr,
                                                     float *kth_out)
{
    constexpr int R = (BOX - 1) / 2, N = R + 1;
    static_assert(!ET || COMBINE_REG, "ET needs the register combiner");
    const int n = P->n_sel;
    const int m = min(n, P->n_best);
    ViewCombiner<COMBINE_REG> comb;
    const int grp_lane0 = (int)(threadIdx.x & 63u & ~(unsigned)(kRowGroup - 1));
    const int src_lane = grp_lane0 + (N - 1);
    for (int vb = 0; vb < n; vb += kRowGroup) {
        float Hl[9];
        homography(P->rc.K_inv, P->view[min(vb + row, n - 1)], pl, Hl);
        const int vend = min(vb + kRowGroup, n);
        for (int v = vb; v < vend; v++) {
            float H[9];
#pragma unroll
            for (int k = 0; k < 9; k++) H[k] = __shfl(Hl[k], grp_lane0 + (v - vb));
            const bool safe = window_z_safe(H, (float)(px - R), (float)(px + R), (float)(py - R), (float)(py + R));
            float tau = __builtin_inff();
            if constexpr (ET)
                if (et_on) tau = __builtin_fminf(comb.kth(m), thr);
            float c;
            if (__all(safe))
                c = view_cost_rows<BOX, true, ET>(P, P->view[v], H, tp0, tws, lut, px, py, row, tau);
            else
                c = view_cost_rows<BOX, false, ET>(P, P->view[v], H, tp0, tws, lut, px, py, row, tau);
            comb.add(__shfl(c, src_lane), v, cv);
        }
    }
    if constexpr (ET)
        if (kth_out) *kth_out = comb.kth(m);
    return comb.finish(P, n, cv);
}

// One colour of one iteration: propagation as in sweep_kernel (one lane per pixel), refinement by
// groups of 8 lanes as described at the top.  Gray packed planes (both layouts) with float-encoded
// offsets, compile-time box <= 15, register combiner.
template <int BOX>
__global__ __launch_bounds__(kThreads, 4) void sweep_rows_kernel(const Problem *__restrict__ P,
                                                                 float4 *__restrict__ norm4,
                                                                 float *__restrict__ cost, int colour,
                                                                 uint32_t phase, unsigned stages, unsigned tune)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CH = 1;
    const Win<BOX> win(P);
    const int rows = P->rows, cols = P->cols;
    SweepLane L;
    sweep_setup<BOX, CH, kRowsTilePad>(L, P, lds, norm4, cost, colour, stages, tune, true);
    RefineDraws R;
    refine_init(R, P, stages);
    const bool et_on = P->et_enable && !(tune & Tune::kNoEarlyExit);

    // ---- propagation rounds: compacted (pixel, candidate) tasks, one per lane ----
    const int prop_rounds = (L.n_tasks + kThreads - 1) / kThreads;
    for (int r = 0; r < prop_rounds; r++) {
        const int pos = r * kThreads + threadIdx.x;
        if (pos < L.n_tasks) {
            const unsigned t = L.btask[pos];
            const int owner = (int)(t & 255u), slot = (int)(t >> 8);
            int olx, oly;
            owner_pixel(L, owner, colour, olx, oly);
            const int epx = L.x0 + olx, epy = L.y0 + oly;
            int nb;
            neighbour(slot, epx, epy, rows, cols, epy * cols + epx, nb);
            const float4 cand = norm4[nb];
            const float *etp0 = L.tile + ((oly + L.hh) * L.tw + (olx + L.hw)) * 4;
            // (value-exact rule of multiview_cost only: these costs are stored)
            L.bres[slot * kThreads + owner] = multiview_cost<BOX, true, true, true, CH, true>(
                P, etp0, L.tw, lds, L.cv, epx, epy, cand, win, et_on, __builtin_inff(), nullptr);
        }
    }
    __syncthreads();
    sweep_replay(L, P, norm4);
    refine_begin(R, L, P, phase);

    // ---- refinement steps: the owner draws its candidate, groups evaluate all 256, the owner accepts ----
    const int grp = threadIdx.x / kRowGroup, row = threadIdx.x % kRowGroup;
    float4 *candbuf = reinterpret_cast<float4 *>(L.btask);  // the task list is dead now (same 4 KB)
    float *cres = L.bres;                  // [256] result per owner
    float *cbound = L.bres + kThreads;     // [256] the cost each candidate has to beat
    for (int step = 0; step < R.nref; step++) {
        float4 cand = make_float4(0.f, 0.f, -1.f, 1.f);
        float d_new = 0.f;
        if (L.active) cand = refine_candidate(R, L, P, d_new);
        refine_next_step(R);
        __syncthreads();  // the previous step's reads of cres / candbuf are done
        candbuf[threadIdx.x] = cand;
        cbound[threadIdx.x] = L.active ? L.cst : -1.0f;  // < 0: nothing can be accepted, cut off at once
        __syncthreads();
        const float theta = step < 3 ? P->et_theta[step] : __builtin_inff();  // (as in sweep_kernel)
        for (int r = 0; r < kThreads / kRowTasks; r++) {
            const int owner = r * kRowTasks + grp;
            int olx, oly;
            owner_pixel(L, owner, colour, olx, oly);
            // pixels outside the image (ragged last tile) evaluate their dummy plane at the clamped
            // position: harmless, never read back
            const int epx = min(L.x0 + olx, cols - 1), epy = min(L.y0 + oly, rows - 1);
            const float4 ecand = candbuf[owner];
            const float bound = cbound[owner];
            const float *etp0 = L.tile + (((epy - L.y0) + L.hh) * L.tw + ((epx - L.x0) + L.hw)) * 4;
            // bound the evaluation by theta * bound; redo the groups whose outcome that leaves open
            float thr = et_on && step < 3 ? theta * bound : __builtin_inff();
            float c = 0.0f;
            bool need = true;
            for (int pass = 0; pass < 2; pass++) {
                if (need) {
                    float kth;
                    const float cc = multiview_cost_rows<BOX, true, true>(P, etp0, L.tw, lds, L.cv, epx, epy, ecand,
                                                                          row, et_on, thr, &kth);
                    const bool open = kth >= thr && cc < bound;
                    if (open) {
                        thr = __builtin_inff();
                    } else {
                        c = cc;
                        need = false;
                    }
                }
                if (!__any(need)) break;
            }
            if (row == 0) cres[owner] = c;
        }
        __syncthreads();
        if (L.active) {
            const float c = cres[threadIdx.x];
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

}  // namespace pm
