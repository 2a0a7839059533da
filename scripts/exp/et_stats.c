/*
 * et_stats.c -- EXPERIMENT (CPU, test infrastructure): how much patch-cost work could be removed
 * exactly on a config-C-like workload?  Includes the oracle restatement and replays the red-black
 * schedule on a window of the frame with the kernels' skip rules (A)/(D)/(H), recording for every
 * evaluated (pixel, candidate):
 *   - per-view column-prefix costs -> simulated early-termination policies (lane and wave level)
 *   - the (source pixel, orientation) groups of propagation candidates -> dis-sharing potential
 * Built and driven by scripts/exp/et_stats.py.  Nothing here is linked into the product.
 */
#include "../../oracle/gipuma_oracle.c"

#include <stdio.h>

#define MAXC 16
#ifndef THETA1
#define THETA1 1.0f
#define THETA2 1.5f
#define THETA3 2.0f
#endif /* window columns (box <= 31) */

/* per-view prefix: pre[c] = cost after window column c, reference order (go_view_cost) */
static float *g_terms = NULL, *g_wts = NULL; /* optional per-sample outputs of view_prefix: w*dis, w */
#pragma omp threadprivate(g_terms, g_wts)
static int view_prefix(const gipuma_hip_desc *d, int view, int px, int py, const float pl[4], float *pre)
{
    int ns = 0;
    const gipuma_hip_params *ap = &d->params;
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *ref = d->images[0];
    const float *src = d->images[view];
    const int hRad = (ap->box_hsize - 1) / 2;
    const int vRad = (ap->box_vsize - 1) / 2;
    const float alpha = ap->alpha, tau_color = ap->tau_color, tau_gradient = ap->tau_gradient;
    const float gamma = ap->gamma;
    const float oma = 1.f - alpha;
    float H[9];
    go_homography(&d->cameras[0], &d->cameras[view], pl, pl[3], H);
    const float centre = go_texel(ref, rows, cols, pitch, px, py);
    float cost = 0.0f;
    int nc = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        const float qx = (float)(px + i);
        const float X0 = fmaf(H[0], qx, H[2]);
        const float Y0 = fmaf(H[3], qx, H[5]);
        const float Z0 = fmaf(H[6], qx, H[8]);
        for (int j = -vRad; j < vRad + 1; j += GO_WIN_INCREMENT) {
            const int ix = px + i, iy = py + j;
            const float qy = (float)iy;
            const float leftValue = go_texel(ref, rows, cols, pitch, ix, iy);
            const float colorDis = fabsf(leftValue - centre);
            const float w = go_exp(-colorDis / gamma);
            const float X = fmaf(H[1], qy, X0);
            const float Y = fmaf(H[4], qy, Y0);
            const float Z = fmaf(H[7], qy, Z0);
            const float rz = 1.0f / Z;
            const float sx = X * rz, sy = Y * rz;
            float s[5];
            go_sample5(src, rows, cols, pitch, sx, sy, s);
            const float gx2 = s[1] - s[2];
            const float gy2 = s[3] - s[4];
            const float colDiff = fabsf(leftValue - s[0]);
            const float up = go_texel(ref, rows, cols, pitch, ix, iy - 1);
            const float down = go_texel(ref, rows, cols, pitch, ix, iy + 1);
            const float left = go_texel(ref, rows, cols, pitch, ix - 1, iy);
            const float right = go_texel(ref, rows, cols, pitch, ix + 1, iy);
            const float gradX = (right - left) - gx2;
            const float gradY = (down - up) - gy2;
            const float gradDis = fminf((fabsf(gradX) + fabsf(gradY)) * 0.0625f, tau_gradient);
            const float colDis = fminf(colDiff, tau_color);
            const float dis = fmaf(alpha, gradDis, oma * colDis);
            cost = fmaf(w, dis, cost);
            if (g_terms) { g_terms[ns] = w * dis; g_wts[ns] = w; }
            ns++;
        }
        pre[nc++] = cost;
    }
    return nc;
}

/* one evaluated task: prefixes of all views; returns the exact aggregated cost */
typedef struct {
    float pre[GIPUMA_HIP_MAX_VIEWS][MAXC];
    int nv, nc;
    float F; /* exact multi-view cost */
} task_eval;

static void eval_task(const gipuma_hip_desc *d, int x, int y, const float pl[4], task_eval *t)
{
    float cv[GIPUMA_HIP_MAX_VIEWS];
    t->nv = d->n_selected;
    for (int i = 0; i < t->nv; i++) {
        t->nc = view_prefix(d, d->selected[i], x, y, pl, t->pre[i]);
        cv[i] = t->pre[i][t->nc - 1];
    }
    t->F = go_aggregate(cv, t->nv, d->params.cost_comb, d->params.n_best, d->params.good_factor);
}

/* Early-termination policy, sequential over views in `order`; a view stops after the first
 * column c with prefix >= min(b_{m-1}, kappa*m*B) (b: m smallest lower bounds so far).
 * stop[k] = columns evaluated for the k-th processed view.  Returns decision (1 = accept)
 * and checks it against the exact one. */
static int g_use_kth = 1; /* 0: a view is cut off at thr only (items independent of the other views) */
#pragma omp threadprivate(g_use_kth)
static int policy_run(const task_eval *t, const int *order, int m, float B, float kappa, unsigned char *stop,
                      int *ambiguous)
{
    float b[8];
    int trunc_in[8];
    for (int i = 0; i < m; i++) {
        b[i] = INFINITY;
        trunc_in[i] = 0;
    }
    const float thr = kappa * (float)m * B * 1.000001f;
    for (int k = 0; k < t->nv; k++) {
        const int v = order[k];
        const float tau = g_use_kth ? fminf(b[m - 1], thr) : thr;
        int c = 0;
        float p = 0.f;
        int truncated = 0;
        for (; c < t->nc; c++) {
            p = t->pre[v][c];
            if (p >= tau && c < t->nc - 1) {
                truncated = 1;
                c++;
                break;
            }
        }
        stop[k] = (unsigned char)c;
        /* insert p */
        int tr = truncated;
        for (int i = 0; i < m; i++) {
            if (p < b[i]) {
                const float tf = b[i];
                const int tt = trunc_in[i];
                b[i] = p;
                trunc_in[i] = tr;
                p = tf;
                tr = tt;
            }
        }
    }
    float sum = 0.f;
    int anytr = 0;
    for (int i = 0; i < m; i++) {
        sum = sum + b[i];
        anytr |= trunc_in[i];
    }
    const float Fp = sum / (float)m;
    *ambiguous = (Fp < B) && anytr;
    return Fp < B;
}

typedef struct {
    /* per launch */
    double prop_tasks, ref_tasks[4];
    double cols_full_prop, cols_full_ref[4];
    /* lane-level evaluated columns by policy p (0: fixed order k=1, 1: best-first per lane k=1,
     * 2: fixed order kappa = 1/m i.e. thr = B (ambiguity counted), 3: wave-adaptive order) */
    double cols_lane_prop[4], cols_lane_ref[4][4];
    double cols_wave_ref[4][4]; /* wave-level (max over the 64 lanes per processed view) */
    double cols_wave_prop[4];
    double ambiguous_prop[4], ambiguous_ref[4][4];
    double wrong[4];
    /* sharing */
    double jobs, job_union_cols, job_target_cols, job_hist[5];
    double accepted_prop, accepted_ref[4];
    double ratio_hist_ref[4][8]; /* F/B histogram per refine step: <1, <1.2, <1.5, <2, <3, <5, <10, >= */
    double ratio_hist_prop[8];
    double plane_groups, plane_union_samples, plane_task_samples, plane_bbox_samples, plane_maxgroup;
    double cols_w8_ref[4][4]; /* like cols_wave_ref but a 'wave' of 8 tasks (8 lanes per task) */
    double seen4, seen8, seen32;  /* needed prop tasks whose plane this pixel evaluated before (ring of K) */
    double seen_own8, seen_own16;
    /* runs of tasks with the same plane on the same row, x stepping by 2 (xrun) / same column, y stepping by 2
     * (yrun): window columns (rows) evaluated = 7 + length instead of 8 * length; [0] = sum of (7 + k),
     * [1] = sum of 8 k, [2] = number of runs, [3] = runs of length 1 */
    double xrun[4], yrun[4]; /* ... with the pixel's own refinement-accepted planes in a second ring of 8 / 16 as well */
    double cols_sorted_ref[4][4]; /* wave-level with lanes regrouped by a predicted stop column */
    /* workgroup pool: per view, rounds of G columns (G = 1, 2, 4); after a round the lanes that reached
     * their bound drop out and the survivors are compacted into ceil(alive/64) wavefronts; lanes left
     * ambiguous are redone in full (compacted too) */
    double cols_pool_ref[4][4][3];
    double cols_wave_nr[4][4]; /* wave-level without the redo cost */
    /* two-phase: every lane does the first G0 = 1, 2, 3, 4 columns of every view of its own task, the
     * surviving (task, view) items of [half the views | all views] are compacted into wavefronts of 64
     * that run to their slowest item's stop (theta 1.0, thr only); [step][G0-1][half/all] */
    double cols_two_phase[4][4][2];
    double items_alive[4][4]; /* surviving items after G0 columns / all items */
    /* propagation tasks bounded item-wise against the pixel's cost at the start of the half-sweep:
     * [0] thr only, open tasks (F' < B0 with a truncated view) completed in full; [1] the same with
     * min(k-th smallest, thr) (sequential views); [2] fraction of tasks left open under [0]; [3] under [1] */
    double prop_item[4];
    /* refinement lower-bound prefilter: items (candidate, view) whose sum over the K highest-weight samples
     * (K = 4, 8, 12, 16, 24, 32) already reaches thr = B; [step][K index]; lb_fixed: the 16 samples of the
     * centre 4x4 / the 4 of the centre 2x2; lb_items: all items; lb_cand_dead[step][K]: candidates whose
     * F' (mean of m smallest bounds) >= B from the prefilter alone */
    double lb_dead[4][6], lb_fixed[4][2], lb_items[4], lb_cand_dead[4][6], lb_cands[4];
    /* samples an item needs in weight order until its partial sum reaches thr (64 if never): sum, [step] */
    double lb_need[4];
    /* the same with the samples taken as horizontally adjacent PAIRS (window columns 2c, 2c+1 of a row; the two
     * windows of a pair lie in one cache line) / as QUADS (columns 4c..4c+3), heaviest first: [step][K index] */
    double lb_pair_dead[4][6], lb_quad_dead[4][6];
} launch_stats;

static int ratio_bin(float F, float B)
{
    const float r = F / B;
    if (r < 1.f) return 0;
    if (r < 1.2f) return 1;
    if (r < 1.5f) return 2;
    if (r < 2.f) return 3;
    if (r < 3.f) return 4;
    if (r < 5.f) return 5;
    if (r < 10.f) return 6;
    return 7;
}

static int nb_of(int k, int x, int y, int rows, int cols, int *nx, int *ny)
{
    const int dist = k < 4 ? 1 : 5;
    *nx = x;
    *ny = y;
    switch (k & 3) {
    case 0: *ny = y - dist; return y > dist - 1;
    case 1: *ny = y + dist; return y < rows - dist;
    case 2: *nx = x - dist; return x > dist - 1;
    default: *nx = x + dist; return x < cols - dist;
    }
}

/* window [x0,x1) x [y0,y1) of the frame is swept (tile aligned); statistics over all of it.
 * n_launch half-sweeps; out: n_launch records. */
int et_stats_run(const gipuma_hip_desc *d, int x0, int y0, int x1, int y1, int n_launch, launch_stats *out,
                 int verbose)
{
    const int rows = d->rows, cols = d->cols;
    const size_t np = (size_t)rows * cols;
    float *norm4 = (float *)calloc(np * 4, sizeof(float));
    float *cost = (float *)calloc(np, sizeof(float));
    float *hist = (float *)calloc(np * 4 * 32, sizeof(float)); /* ring of 32 planes per pixel */
    unsigned char *hcount = (unsigned char *)calloc(np, 1);
    unsigned *hpos = (unsigned *)calloc(np, sizeof(unsigned));
    float *hist2 = (float *)calloc(np * 4 * 16, sizeof(float)); /* ring of 16: propagation-evaluated + own refined planes */
    unsigned *hpos2 = (unsigned *)calloc(np, sizeof(unsigned));
    unsigned char *changed = (unsigned char *)malloc(np);
    memset(changed, 1, np);
    const int m = d->params.n_best < d->n_selected ? d->params.n_best : d->n_selected;
    const int nv = d->n_selected;
    /* init on the window + reach */
    const int ya = y0 - 5 < 0 ? 0 : y0 - 5, yb = y1 + 5 > rows ? rows : y1 + 5;
    const int xa = x0 - 5 < 0 ? 0 : x0 - 5, xb = x1 + 5 > cols ? cols : x1 + 5;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = ya; y < yb; y++)
        for (int x = xa; x < xb; x++) go_init_pixel(d, x, y, norm4, cost);
    memset(out, 0, sizeof(launch_stats) * (size_t)n_launch);
    const int tw = 32, th = 16;
    const int ntx = (x1 - x0) / tw, nty = (y1 - y0) / th;
    for (int L = 0; L < n_launch; L++) {
        const int it = L / 2, colour = L & 1;
        const uint32_t phase = go_phase(it, colour);
        const int history = L >= 2;
        launch_stats *S = &out[L];
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int ty = 0; ty < nty; ty++)
            for (int tx = 0; tx < ntx; tx++) {
                launch_stats loc;
                memset(&loc, 0, sizeof loc);
                /* per tile: 256 pixels of the colour, lane id = (ly << 4) | (lx >> 1) */
                static __thread unsigned char stopbuf[4][4][256][GIPUMA_HIP_MAX_VIEWS];
                static __thread unsigned char refvalid[256];
                static __thread unsigned char ambbuf[4][4][256];
                static __thread float keybuf[3][256];
                static __thread unsigned char pstop[4][2048][GIPUMA_HIP_MAX_VIEWS];
                int nprop = 0;
                static __thread struct { int lx, ly; uint32_t pl[4]; } ptask[2048];
                int nptask = 0;
                /* sharing: target masks per (source pixel in extended tile, orientation) */
                static __thread unsigned char jobmask[2][(16 + 10) * (32 + 10)];
                memset(jobmask, 0, sizeof jobmask);
                memset(refvalid, 0, sizeof refvalid);
                double viewcost_sum[GIPUMA_HIP_MAX_VIEWS];
                for (int v = 0; v < nv; v++) viewcost_sum[v] = 0;
                int wave_order[GIPUMA_HIP_MAX_VIEWS];
                static __thread float curcv[256][GIPUMA_HIP_MAX_VIEWS];
                /* pre-pass: view costs of every pixel's current plane (order heuristics) */
                for (int ly = 0; ly < th; ly++)
                    for (int lxh = 0; lxh < 16; lxh++) {
                        const int lx = 2 * lxh + ((ly + colour) & 1);
                        const int x = x0 + tx * tw + lx, y = y0 + ty * th + ly;
                        const int lane = (ly << 4) | lxh;
                        task_eval cur;
                        eval_task(d, x, y, norm4 + 4 * ((size_t)y * cols + x), &cur);
                        for (int v = 0; v < nv; v++) {
                            curcv[lane][v] = cur.pre[v][cur.nc - 1];
                            viewcost_sum[v] += curcv[lane][v];
                        }
                    }
                for (int v = 0; v < nv; v++) wave_order[v] = v;
                for (int a = 1; a < nv; a++) {
                    const int ov = wave_order[a];
                    int bpos = a;
                    for (; bpos >= 1 && viewcost_sum[ov] < viewcost_sum[wave_order[bpos - 1]]; bpos--)
                        wave_order[bpos] = wave_order[bpos - 1];
                    wave_order[bpos] = ov;
                }
                for (int ly = 0; ly < th; ly++)
                    for (int lxh = 0; lxh < 16; lxh++) {
                        const int lx = 2 * lxh + ((ly + colour) & 1);
                        const int x = x0 + tx * tw + lx, y = y0 + ty * th + ly;
                        const int lane = (ly << 4) | lxh;
                        const size_t center = (size_t)y * cols + x;
                        go_pixel_state st;
                        memcpy(st.pl, norm4 + 4 * center, sizeof st.pl);
                        st.cost = cost[center];
                        st.depth = go_depth_from_plane(&d->cameras[0], st.pl, x, y);
                        const float B0 = st.cost;
                        int chg = 0;
                        /* candidates + skip rules */
                        float cands[8][4];
                        int valid[8], need[8];
                        for (int k = 0; k < 8; k++) {
                            int nx, ny;
                            valid[k] = nb_of(k, x, y, rows, cols, &nx, &ny);
                            need[k] = 0;
                            if (!valid[k]) continue;
                            const size_t nb = (size_t)ny * cols + nx;
                            memcpy(cands[k], norm4 + 4 * nb, 16);
                            int fresh = 1;
                            if (!memcmp(cands[k], st.pl, 16)) fresh = 0;
                            if (fresh && history && !changed[nb]) fresh = 0;
                            for (int j = 0; j < k && fresh; j++)
                                if (valid[j] && !memcmp(cands[k], cands[j], 16)) fresh = 0;
                            need[k] = fresh;
                        }
                        /* best-first order of this lane: by the view costs of its current plane */
                        int order_fixed[GIPUMA_HIP_MAX_VIEWS], order_best[GIPUMA_HIP_MAX_VIEWS];
                        {
                            const float *cv = curcv[lane];
                            for (int v = 0; v < nv; v++) {
                                order_fixed[v] = v;
                                order_best[v] = v;
                            }
                            for (int a = 1; a < nv; a++) {
                                const int ov = order_best[a];
                                int bpos = a;
                                for (; bpos >= 1 && cv[ov] < cv[order_best[bpos - 1]]; bpos--)
                                    order_best[bpos] = order_best[bpos - 1];
                                order_best[bpos] = ov;
                            }
                        }
                        for (int k = 0; k < 8; k++) {
                            if (!need[k]) continue;
                            {
                                const float *hp = hist + center * 128;
                                const unsigned n = hpos[center];
                                for (unsigned a = 0; a < 32 && a < n; a++) {
                                    const unsigned idx = (n - 1 - a) & 31;
                                    if (!memcmp(hp + 4 * idx, cands[k], 16)) {
                                        if (a < 4) loc.seen4 += 1;
                                        if (a < 8) loc.seen8 += 1;
                                        loc.seen32 += 1;
                                        break;
                                    }
                                }
                                memcpy(hist + center * 128 + 4 * (hpos[center] & 31), cands[k], 16);
                                hpos[center]++;
                                {
                                    const float *hp2 = hist2 + center * 64;
                                    const unsigned n2 = hpos2[center];
                                    for (unsigned a = 0; a < 16 && a < n2; a++) {
                                        const unsigned idx = (n2 - 1 - a) & 15;
                                        if (!memcmp(hp2 + 4 * idx, cands[k], 16)) {
                                            if (a < 8) loc.seen_own8 += 1;
                                            loc.seen_own16 += 1;
                                            break;
                                        }
                                    }
                                    memcpy(hist2 + center * 64 + 4 * (hpos2[center] & 15), cands[k], 16);
                                    hpos2[center]++;
                                }
                            }
                            task_eval te;
                            eval_task(d, x, y, cands[k], &te);
                            loc.prop_tasks += 1;
                            if (nptask < 2048) { ptask[nptask].lx = lx; ptask[nptask].ly = ly; memcpy(ptask[nptask].pl, cands[k], 16); nptask++; }
                            loc.cols_full_prop += (double)te.nv * te.nc;
                            loc.ratio_hist_prop[ratio_bin(te.F, B0)] += 1;
                            unsigned char stop[GIPUMA_HIP_MAX_VIEWS];
                            int amb;
                            const int *orders[4] = {order_fixed, order_fixed, order_fixed, order_fixed};
                            const float kap[4] = {1.f, THETA1 / (float)m, THETA2 / (float)m, THETA3 / (float)m};
                            for (int p = 0; p < 4; p++) {
                                const int dec = policy_run(&te, orders[p], m, B0, kap[p], stop, &amb);
                                int s = 0;
                                for (int v = 0; v < nv; v++) s += stop[v];
                                loc.cols_lane_prop[p] += s;
                                loc.ambiguous_prop[p] += amb;
                                if (!amb && dec != (te.F < B0)) loc.wrong[p] += 1;
                                if (nprop < 2048) memcpy(pstop[p][nprop], stop, (size_t)nv);
                            }
                            if (nprop < 2048) nprop++;
                            for (int q = 0; q < 2; q++) {
                                g_use_kth = q;
                                const int dec = policy_run(&te, order_fixed, m, B0, THETA1 / (float)m, stop, &amb);
                                g_use_kth = 1;
                                (void)dec;
                                int s2 = 0;
                                for (int v = 0; v < nv; v++) s2 += stop[v];
                                loc.prop_item[q] += amb ? (double)te.nv * te.nc : (double)s2;
                                loc.prop_item[2 + q] += amb;
                            }
                            /* sharing bookkeeping */
                            {
                                const int dist = k < 4 ? 1 : 5;
                                const int sx = lx + ((k & 3) == 2 ? -dist : (k & 3) == 3 ? dist : 0) + 5;
                                const int sy = ly + ((k & 3) == 0 ? -dist : (k & 3) == 1 ? dist : 0) + 5;
                                const int orient = (k & 3) >= 2 ? 0 : 1; /* 0: horizontal targets */
                                /* target index 0..3: offsets -5,-1,+1,+5 of the target from the source */
                                const int off = (k & 3) == 2 || (k & 3) == 0 ? dist : -dist; /* target - source */
                                const int ti = off == -5 ? 0 : off == -1 ? 1 : off == 1 ? 2 : 3;
                                jobmask[orient][sy * 42 + sx] |= (unsigned char)(1u << ti);
                            }
                            /* accept replay */
                            const float dnew = go_depth_from_plane(&d->cameras[0], cands[k], x, y);
                            if (dnew >= d->cameras[0].depth_min && dnew <= d->cameras[0].depth_max && te.F < st.cost) {
                                st.depth = dnew;
                                memcpy(st.pl, cands[k], 16);
                                st.cost = te.F;
                                chg = 1;
                                loc.accepted_prop += 1;
                            }
                        }
                        /* refinement */
                        st.depth = go_depth_from_plane(&d->cameras[0], st.pl, x, y);
                        {
                            const gipuma_hip_camera *cam = &d->cameras[0];
                            const gipuma_hip_params *ap = &d->params;
                            float view[3];
                            go_view_vector(cam, x, y, view);
                            float deltaN = 1.0f;
                            uint32_t draw = 0;
                            int step = 0;
                            for (float deltaZ = ap->max_disparity / 2.0f; deltaZ >= 0.01f; deltaZ = deltaZ / 10.0f, step++) {
                                const float disp = go_disp_depth(cam->f, cam->baseline, st.depth);
                                const float minDelta = -fminf(deltaZ, ap->min_disparity + disp);
                                const float maxDelta = fminf(deltaZ, ap->max_disparity - disp);
                                const float u0 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
                                const float u1 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
                                const float u2 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
                                const float u3 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
                                const float dz = go_between(u0, minDelta, maxDelta);
                                float dispOut = fminf(fmaxf(disp + dz, ap->min_disparity), ap->max_disparity);
                                const float depthOut = go_disp_depth(cam->f, cam->baseline, dispOut);
                                float cand[4];
                                cand[0] = st.pl[0] + go_between(u1, -deltaN, deltaN);
                                cand[1] = st.pl[1] + go_between(u2, -deltaN, deltaN);
                                cand[2] = st.pl[2] + go_between(u3, -deltaN, deltaN);
                                go_normalize(cand);
                                go_on_hemisphere(cand, view);
                                cand[3] = go_plane_d(cam, cand, x, y, depthOut);
                                task_eval te;
                                eval_task(d, x, y, cand, &te);
                                const int sidx = step < 3 ? step : 3;
                                if (step < 3 && te.nc * te.nc <= 64) {
                                    static const int KS[6] = {4, 8, 12, 16, 24, 32};
                                    float terms[64], wts[64];
                                    int ord[64], pord[64], qord[64];
                                    const int ns = te.nc * te.nc;
                                    float lbv[6][GIPUMA_HIP_MAX_VIEWS];
                                    for (int v = 0; v < nv; v++) {
                                        float dummy[MAXC];
                                        g_terms = terms;
                                        g_wts = wts;
                                        view_prefix(d, d->selected[v], x, y, cand, dummy);
                                        g_terms = NULL;
                                        if (v == 0) {
                                            for (int a = 0; a < ns; a++) ord[a] = a;
                                            for (int a = 1; a < ns; a++) {
                                                const int o = ord[a];
                                                int b = a;
                                                for (; b >= 1 && wts[o] > wts[ord[b - 1]]; b--) ord[b] = ord[b - 1];
                                                ord[b] = o;
                                            }
                                        }
                                        if (v == 0) {
                                            /* sample index = col * nc + row; pairs: cols (2c, 2c+1) of a row, by summed weight */
                                            for (int grp = 2; grp <= 4; grp += 2) {
                                                int *o = grp == 2 ? pord : qord;
                                                const int ng = te.nc / grp;
                                                int ids[64]; float gw[64]; int n = 0;
                                                for (int ri = 0; ri < te.nc; ri++)
                                                    for (int c = 0; c < ng; c++) {
                                                        float wsum = 0.f;
                                                        for (int e = 0; e < grp; e++) wsum += wts[(grp * c + e) * te.nc + ri];
                                                        ids[n] = c * 64 + ri; gw[n] = wsum; n++;
                                                    }
                                                for (int a = 1; a < n; a++) {
                                                    const int oi = ids[a]; const float ow = gw[a];
                                                    int b = a;
                                                    for (; b >= 1 && ow > gw[b - 1]; b--) { ids[b] = ids[b - 1]; gw[b] = gw[b - 1]; }
                                                    ids[b] = oi; gw[b] = ow;
                                                }
                                                int k = 0;
                                                for (int a = 0; a < n && k + grp <= 64; a++)
                                                    for (int e = 0; e < grp; e++) o[k++] = (grp * (ids[a] / 64) + e) * te.nc + (ids[a] % 64);
                                                for (; k < 64; k++) o[k] = 0;
                                            }
                                        }
                                        for (int mode = 0; mode < 2; mode++) {
                                            const int *o = mode == 0 ? pord : qord;
                                            float acc2 = 0.f;
                                            int ki2 = 0;
                                            for (int a = 0; a < 32 && a < ns; a++) {
                                                acc2 += terms[o[a]];
                                                while (ki2 < 6 && KS[ki2] == a + 1) {
                                                    if (acc2 * 0.99998f >= st.cost) (mode == 0 ? loc.lb_pair_dead : loc.lb_quad_dead)[step][ki2] += 1;
                                                    ki2++;
                                                }
                                            }
                                        }
                                        const float thrv = st.cost;
                                        float acc = 0.f;
                                        int need = 64, ki = 0;
                                        for (int a = 0; a < ns; a++) {
                                            acc += terms[ord[a]];
                                            if (need == 64 && acc * 0.99998f >= thrv) need = a + 1;
                                            while (ki < 6 && KS[ki] == a + 1) {
                                                lbv[ki][v] = acc * 0.99998f;
                                                if (acc * 0.99998f >= thrv) loc.lb_dead[step][ki] += 1;
                                                ki++;
                                            }
                                        }
                                        loc.lb_need[step] += need;
                                        loc.lb_items[step] += 1;
                                        /* fixed patterns: sample index = col * nc + row; centre 4x4 = cols/rows nc/2-2 .. nc/2+1 */
                                        float f16 = 0.f, f4 = 0.f;
                                        for (int ci = te.nc / 2 - 2; ci < te.nc / 2 + 2; ci++)
                                            for (int ri = te.nc / 2 - 2; ri < te.nc / 2 + 2; ri++) {
                                                f16 += terms[ci * te.nc + ri];
                                                if (ci >= te.nc / 2 - 1 && ci <= te.nc / 2 && ri >= te.nc / 2 - 1 && ri <= te.nc / 2) f4 += terms[ci * te.nc + ri];
                                            }
                                        if (f16 * 0.99998f >= thrv) loc.lb_fixed[step][0] += 1;
                                        if (f4 * 0.99998f >= thrv) loc.lb_fixed[step][1] += 1;
                                    }
                                    loc.lb_cands[step] += 1;
                                    for (int ki = 0; ki < 6; ki++) {
                                        /* mean of the m smallest bounds */
                                        float b[GIPUMA_HIP_MAX_VIEWS];
                                        for (int v = 0; v < nv; v++) b[v] = lbv[ki][v];
                                        for (int a = 1; a < nv; a++) {
                                            const float o = b[a];
                                            int q = a;
                                            for (; q >= 1 && o < b[q - 1]; q--) b[q] = b[q - 1];
                                            b[q] = o;
                                        }
                                        float sm = 0.f;
                                        for (int a = 0; a < m; a++) sm += b[a];
                                        if (sm / (float)m >= st.cost) loc.lb_cand_dead[step][ki] += 1;
                                    }
                                }
                                if (step < 3) {
                                    /* predicted stop: saturated dis times the running sum of support weights */
                                    const float dismax = (1.f - ap->alpha) * ap->tau_color + ap->alpha * ap->tau_gradient;
                                    const float centre = go_texel(d->images[0], rows, cols, d->pitch, x, y);
                                    const int hR = (ap->box_hsize - 1) / 2, vR = (ap->box_vsize - 1) / 2;
                                    float wsum = 0.f, key = 99.f;
                                    int cc = 0;
                                    for (int i = -hR; i <= hR; i += 2, cc++) {
                                        for (int j = -vR; j <= vR; j += 2)
                                            wsum += go_exp(-fabsf(go_texel(d->images[0], rows, cols, d->pitch, x + i, y + j) - centre) / ap->gamma);
                                        if (key > 98.f && dismax * wsum * 0.6f >= st.cost) key = (float)cc + st.cost / (dismax * wsum * 0.6f);
                                    }
                                    keybuf[step][lane] = key;
                                }
                                loc.ref_tasks[sidx] += 1;
                                loc.cols_full_ref[sidx] += (double)te.nv * te.nc;
                                loc.ratio_hist_ref[sidx][ratio_bin(te.F, st.cost)] += 1;
                                unsigned char stop[GIPUMA_HIP_MAX_VIEWS];
                                int amb;
                                const int *orders[4] = {order_fixed, order_fixed, order_fixed, order_fixed};
                                const float kap[4] = {1.f, THETA1 / (float)m, THETA2 / (float)m, THETA3 / (float)m};
                                for (int p = 0; p < 4; p++) {
                                    g_use_kth = p != 3; /* refinement, policy 3: theta 1.0, thr only */
                                    const int dec = policy_run(&te, orders[p], m, st.cost, p == 3 ? THETA1 / (float)m : kap[p], stop, &amb);
                                    g_use_kth = 1;
                                    int s = 0;
                                    for (int v = 0; v < nv; v++) s += stop[v];
                                    loc.cols_lane_ref[p][sidx] += s;
                                    loc.ambiguous_ref[p][sidx] += amb;
                                    if (!amb && dec != (te.F < st.cost)) loc.wrong[p] += 1;
                                    if (step < 4) { memcpy(stopbuf[p][step][lane], stop, (size_t)nv); ambbuf[p][step][lane] = (unsigned char)amb; }
                                }
                                refvalid[lane] = 1;
                                if (te.F < st.cost) {
                                    memcpy(hist2 + center * 64 + 4 * (hpos2[center] & 15), cand, 16);
                                    hpos2[center]++;
                                    st.cost = te.F;
                                    st.depth = depthOut;
                                    memcpy(st.pl, cand, 16);
                                    chg = 1;
                                    loc.accepted_ref[sidx] += 1;
                                }
                                deltaN = deltaN / 4.0f;
                            }
                        }
                        cost[center] = st.cost;
                        memcpy(norm4 + 4 * center, st.pl, 16);
                        changed[center] = (unsigned char)chg;
                    }
                /* wave level: refinement, wave = 64 consecutive lanes; a wave with an ambiguous lane
                 * re-runs the step under policy 0 (3B) */
                for (int p = 0; p < 4; p++)
                    for (int step = 0; step < 4; step++)
                        for (int w = 0; w < 4; w++)
                            for (int k = 0; k < nv; k++) {
                                int mx = 0;
                                for (int l = 0; l < 64; l++)
                                    if (refvalid[w * 64 + l] && stopbuf[p][step][w * 64 + l][k] > mx) mx = stopbuf[p][step][w * 64 + l][k];
                                loc.cols_wave_nr[p][step] += 64.0 * mx;
                            }
                for (int p = 0; p < 4; p++)
                    for (int step = 0; step < 3; step++)
                        for (int w = 0; w < 4; w++) {
                            int anyamb = 0;
                            for (int l = 0; l < 64; l++)
                                if (refvalid[w * 64 + l] && ambbuf[p][step][w * 64 + l]) anyamb = 1;
                            for (int k = 0; k < nv; k++) {
                                int mx = 0, mx0 = 0;
                                for (int l = 0; l < 64; l++)
                                    if (refvalid[w * 64 + l]) {
                                        if (stopbuf[p][step][w * 64 + l][k] > mx) mx = stopbuf[p][step][w * 64 + l][k];
                                        if (stopbuf[0][step][w * 64 + l][k] > mx0) mx0 = stopbuf[0][step][w * 64 + l][k];
                                    }
                                loc.cols_wave_ref[p][step] += 64.0 * mx + (anyamb ? 64.0 * mx0 : 0.0);
                            }
                        }
                for (int p = 0; p < 4; p++)
                    for (int step = 0; step < 3; step++)
                        for (int w = 0; w < 32; w++) {
                            int anyamb = 0;
                            for (int l = 0; l < 8; l++)
                                if (refvalid[w * 8 + l] && ambbuf[p][step][w * 8 + l]) anyamb = 1;
                            for (int k = 0; k < nv; k++) {
                                int mx = 0, mx0 = 0;
                                for (int l = 0; l < 8; l++)
                                    if (refvalid[w * 8 + l]) {
                                        if (stopbuf[p][step][w * 8 + l][k] > mx) mx = stopbuf[p][step][w * 8 + l][k];
                                        if (stopbuf[0][step][w * 8 + l][k] > mx0) mx0 = stopbuf[0][step][w * 8 + l][k];
                                    }
                                loc.cols_w8_ref[p][step] += 8.0 * mx + (anyamb ? 8.0 * mx0 : 0.0);
                            }
                        }
                for (int p = 0; p < 4; p++)
                    for (int step = 0; step < 4; step++)
                        for (int gi = 0; gi < 3; gi++) {
                            const int G = 1 << gi;
                            const int nc = (d->params.box_hsize + 1) / 2;
                            int namb = 0;
                            for (int l = 0; l < 256; l++)
                                if (refvalid[l] && ambbuf[p][step][l]) namb++;
                            double work = 0;
                            for (int k = 0; k < nv; k++)
                                for (int c0 = 0; c0 < nc; c0 += G) {
                                    int alive = 0;
                                    for (int l = 0; l < 256; l++)
                                        if (refvalid[l] && stopbuf[p][step][l][k] > c0) alive++;
                                    const int g = c0 + G <= nc ? G : nc - c0;
                                    work += (double)((alive + 63) / 64) * 64.0 * g;
                                }
                            work += (double)((namb + 63) / 64) * 64.0 * nc * nv;
                            loc.cols_pool_ref[p][step][gi] += work;
                        }
                for (int step = 0; step < 3; step++)
                    for (int g0 = 1; g0 <= 4; g0++)
                        for (int mode = 0; mode < 2; mode++) {
                            const int p = 3;
                            const int nc = (d->params.box_hsize + 1) / 2;
                            const int grp = mode == 0 ? (nv + 1) / 2 : nv;
                            double work = 0;
                            int nvalid = 0;
                            for (int l = 0; l < 256; l++) nvalid += refvalid[l];
                            work += 256.0 * nv * g0;
                            for (int vb = 0; vb < nv; vb += grp) {
                                int cnt = 0, mx = 0;
                                for (int k = vb; k < vb + grp && k < nv; k++)
                                    for (int l = 0; l < 256; l++) {
                                        if (!refvalid[l] || stopbuf[p][step][l][k] <= g0) continue;
                                        if (mode == 0 && g0 == 1) loc.items_alive[step][0] += 0; /* (counted below) */
                                        if (stopbuf[p][step][l][k] > mx) mx = stopbuf[p][step][l][k];
                                        if (++cnt == 64) {
                                            work += 64.0 * (mx - g0);
                                            cnt = 0;
                                            mx = 0;
                                        }
                                    }
                                if (cnt) work += 64.0 * (mx - g0);
                            }
                            loc.cols_two_phase[step][g0 - 1][mode] += work;
                            if (mode == 1) {
                                int alive = 0;
                                for (int k = 0; k < nv; k++)
                                    for (int l = 0; l < 256; l++)
                                        if (refvalid[l] && stopbuf[p][step][l][k] > g0) alive++;
                                loc.items_alive[step][g0 - 1] += alive;
                            }
                            (void)nc;
                        }
                /* lanes regrouped by the predicted key: rank order -> waves of 64 */
                for (int step = 0; step < 3; step++) {
                    int idx[256];
                    for (int l = 0; l < 256; l++) idx[l] = l;
                    for (int a = 1; a < 256; a++) {
                        const int v = idx[a];
                        int b = a;
                        for (; b >= 1 && keybuf[step][v] < keybuf[step][idx[b - 1]]; b--) idx[b] = idx[b - 1];
                        idx[b] = v;
                    }
                    for (int p = 0; p < 4; p++)
                        for (int w = 0; w < 4; w++) {
                            int anyamb = 0;
                            for (int l = 0; l < 64; l++)
                                if (refvalid[idx[w * 64 + l]] && ambbuf[p][step][idx[w * 64 + l]]) anyamb = 1;
                            for (int k = 0; k < nv; k++) {
                                int mx = 0, mx0 = 0;
                                for (int l = 0; l < 64; l++) {
                                    const int ll = idx[w * 64 + l];
                                    if (!refvalid[ll]) continue;
                                    if (stopbuf[p][step][ll][k] > mx) mx = stopbuf[p][step][ll][k];
                                    if (stopbuf[0][step][ll][k] > mx0) mx0 = stopbuf[0][step][ll][k];
                                }
                                loc.cols_sorted_ref[p][step] += 64.0 * mx + (anyamb ? 64.0 * mx0 : 0.0);
                            }
                        }
                }
                /* propagation: groups of 64 tasks in owner order */
                for (int p = 0; p < 4; p++)
                    for (int g = 0; g < nprop; g += 64)
                        for (int k = 0; k < nv; k++) {
                            int mx = 0;
                            const int e = g + 64 < nprop ? g + 64 : nprop;
                            for (int l = g; l < e; l++)
                                if (pstop[p][l][k] > mx) mx = pstop[p][l][k];
                            loc.cols_wave_prop[p] += 64.0 * mx;
                        }
                /* runs along x / y */
                for (int dir = 0; dir < 2; dir++) {
                    static __thread unsigned char usedr[2048];
                    memset(usedr, 0, (size_t)nptask);
                    double *out4 = dir == 0 ? loc.xrun : loc.yrun;
                    for (int a = 0; a < nptask; a++) {
                        if (usedr[a]) continue;
                        /* find the start of the run containing a, then walk it */
                        int cx = ptask[a].lx, cy = ptask[a].ly;
                        for (;;) {
                            int found = -1;
                            for (int b = 0; b < nptask; b++)
                                if (!usedr[b] && b != a && !memcmp(ptask[a].pl, ptask[b].pl, 16) &&
                                    ptask[b].lx == cx - (dir == 0 ? 2 : 0) && ptask[b].ly == cy - (dir == 0 ? 0 : 2)) { found = b; break; }
                            if (found < 0) break;
                            cx = ptask[found].lx; cy = ptask[found].ly;
                        }
                        int k = 0;
                        for (;;) {
                            int found = -1;
                            for (int b = 0; b < nptask; b++)
                                if (!usedr[b] && !memcmp(ptask[a].pl, ptask[b].pl, 16) && ptask[b].lx == cx && ptask[b].ly == cy) { found = b; break; }
                            if (found < 0) break;
                            usedr[found] = 1;
                            k++;
                            if (dir == 0) cx += 2; else cy += 2;
                        }
                        if (k == 0) { usedr[a] = 1; k = 1; }
                        out4[0] += 7 + k;
                        out4[1] += 8.0 * k;
                        out4[2] += 1;
                        if (k == 1) out4[3] += 1;
                    }
                }
                /* plane-keyed sharing: group tasks by (plane bits, parity class), union of sample points */
                {
                    static __thread unsigned char used[2048];
                    static __thread unsigned char bitmap[(16 + 16) * (32 + 16)];
                    memset(used, 0, (size_t)nptask);
                    for (int a = 0; a < nptask; a++) {
                        if (used[a]) continue;
                        memset(bitmap, 0, sizeof bitmap);
                        int cnt = 0;
                        for (int b = a; b < nptask; b++) {
                            if (used[b] || memcmp(ptask[a].pl, ptask[b].pl, 16)) continue;
                            if (((ptask[a].lx ^ ptask[b].lx) & 1) || ((ptask[a].ly ^ ptask[b].ly) & 1)) continue;
                            used[b] = 1;
                            cnt++;
                            for (int i = -7; i <= 7; i += 2)
                                for (int j = -7; j <= 7; j += 2)
                                    bitmap[(ptask[b].ly + j + 8) * 48 + ptask[b].lx + i + 8] = 1;
                        }
                        int un = 0;
                        int bx0 = 999, bx1 = -999, by0 = 999, by1 = -999;
                        for (size_t c = 0; c < sizeof bitmap; c++) {
                            un += bitmap[c];
                            if (bitmap[c]) {
                                const int by = (int)(c / 48), bx = (int)(c % 48);
                                if (bx < bx0) bx0 = bx;
                                if (bx > bx1) bx1 = bx;
                                if (by < by0) by0 = by;
                                if (by > by1) by1 = by;
                            }
                        }
                        loc.plane_bbox_samples += ((bx1 - bx0) / 2 + 1) * ((by1 - by0) / 2 + 1);
                        if (cnt > loc.plane_maxgroup) loc.plane_maxgroup = cnt;
                        loc.plane_groups += 1;
                        loc.plane_union_samples += un;
                        loc.plane_task_samples += 64.0 * cnt;
                    }
                }
                /* sharing */
                for (int o = 0; o < 2; o++)
                    for (int c = 0; c < 26 * 42; c++) {
                        const unsigned mk = jobmask[o][c];
                        if (!mk) continue;
                        loc.jobs += 1;
                        /* union columns: target ti covers union columns start[ti]..start[ti]+7 */
                        static const int start[4] = {0, 2, 3, 5};
                        int lo = 99, hi = -1, nt = 0;
                        for (int ti = 0; ti < 4; ti++)
                            if (mk & (1u << ti)) {
                                /* target offset order: mask bit = target-source offset -5,-1,+1,+5 */
                                if (start[ti] < lo) lo = start[ti];
                                if (start[ti] + 8 > hi) hi = start[ti] + 8;
                                nt++;
                            }
                        /* exact union (columns covered by at least one target) */
                        int covered = 0;
                        for (int cc = 0; cc < 13; cc++) {
                            int in = 0;
                            for (int ti = 0; ti < 4; ti++)
                                if ((mk & (1u << ti)) && cc >= start[ti] && cc < start[ti] + 8) in = 1;
                            covered += in;
                        }
                        loc.job_union_cols += covered;
                        loc.job_target_cols += 8.0 * nt;
                        loc.job_hist[nt] += 1;
                    }
#pragma omp critical
                {
                    double *a = (double *)S;
                    const double *b = (const double *)&loc;
                    for (size_t i = 0; i < sizeof(launch_stats) / sizeof(double); i++) a[i] += b[i];
                }
            }
        if (verbose) {
            fprintf(stderr, "launch %d done: prop %.0f tasks\n", L, S->prop_tasks);
        }
    }
    free(norm4);
    free(cost);
    free(changed);
    return 0;
}

int et_stats_sizeof(void) { return (int)sizeof(launch_stats); }
