/*
 * gipuma_oracle.c -- CPU restatement of Gipuma's red-black PatchMatch loop (reference
 * gipuma.cu).  TEST INFRASTRUCTURE ONLY: see gipuma_oracle.h for who may load it and for the
 * pin status.
 *
 * Plain C99, fp32 throughout, built with -ffp-contract=off so the only fused multiply-adds
 * are the ones spelled fmaf().  Each function cites the reference lines it restates.
 *
 * Numerical model (where the reference's own arithmetic is not reproducible on any CPU, the
 * model is stated here once and used by BOTH this oracle and the HIP kernels):
 *   M1  bilinear sampling: the reference samples float textures with cudaFilterModeLinear at
 *       (x+0.5, y+0.5) (main.cpp:644-648, gipuma.cu:251-253), i.e. hardware 1.8 fixed-point
 *       weights.  gfx950 has no texture unit; the model is an exact fp32 lerp
 *       t0 + a*(t1-t0) (one fmaf) of the texels floor(x), floor(x)+1 with clamp-to-edge
 *       texel addressing, a = x - floor(x).  The +-1 gradient taps (gipuma.cu:251-252) reuse
 *       the centre tap's a/b and move the texel window by whole texels; since round 5 the two
 *       gradient differences are taken on the texels BEFORE the interpolation (bilinear
 *       interpolation is linear in the texels) and columns are interpolated along y first:
 *       go_taps3_s() below -- 24 instead of 28 operations per sample on the GPU.  Measured against
 *       the reference's own code, which of these orders is used does not matter (flavours below).
 *   M2  expf / rsqrtf / division are approximate in the reference build (--use_fast_math,
 *       CMakeLists.txt:23).  Model: exp is go_exp() below (Cephes-style, ~1 ulp), rsqrtf(x) is
 *       1/sqrtf(x).  x/z and y/z of the warped point: IEEE divisions since round 6 (the source's own
 *       operation; flavour bit 2 clear: x*(1/z), y*(1/z) with an IEEE 1/z, the model of rounds 1-5).
 *   M3  nvcc contracts a*b+c to FMA at will.  Since round 6 the default is NO contraction anywhere (the
 *       source order as g++ compiles it; flavour bit 4 clear: the fmaf nesting rounds 1-5 fixed for the
 *       per-sample loop -- homography application, dis, cost accumulation); the bilinear lerps are fmaf
 *       in every flavour (they stand for the texture unit), everything evaluated once per hypothesis is
 *       unfused, in source order.
 *   M4  random numbers: the reference's cuRAND state is never initialised; the
 *       model is a stateless counter hash keyed by (seed, phase, x, y, draw) -> (0,1].
 */
#include "gipuma_oracle.h"

#include <math.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define GO_HAVE_AVX2 1
#else
#define GO_HAVE_AVX2 0
#endif
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GO_MAXCOST 1000.0f /* config.h:22 */
#define GO_WIN_INCREMENT 2 /* gipuma.cu:28 */

/* ------------------------------------------------------------------------------------------
 * M4: counter-based uniform in (0,1], standing in for curand_uniform (gipuma.cu:138-141)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t go_mix(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x7feb352dU;
    h ^= h >> 15;
    h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}

static inline uint32_t go_rand_u32(uint32_t seed, uint32_t phase, uint32_t x, uint32_t y,
                                   uint32_t draw)
{
    uint32_t h = go_mix(seed + 0x9E3779B9U);
    h = go_mix(h ^ (phase + 0x85EBCA6BU));
    h = go_mix(h ^ (y + 0xC2B2AE35U));
    h = go_mix(h ^ (x + 0x27D4EB2FU));
    h = go_mix(h ^ (draw + 0x165667B1U));
    return h;
}

static inline float go_uniform(uint32_t seed, uint32_t phase, uint32_t x, uint32_t y,
                               uint32_t draw)
{
    const uint32_t h = go_rand_u32(seed, phase, x, y, draw);
    return (float)((h >> 8) + 1U) * 5.9604644775390625e-8f; /* 2^-24: (0,1] exactly */
}

float gipuma_oracle_uniform(uint32_t seed, uint32_t phase, uint32_t x, uint32_t y, uint32_t draw)
{
    return go_uniform(seed, phase, x, y, draw);
}

/* phase ids: init = 0; sweep (iteration, colour) = 1 + 2*iteration + colour */
static inline uint32_t go_phase(int iteration, int colour)
{
    return 1U + 2U * (uint32_t)iteration + (uint32_t)colour;
}

/* curand_between, gipuma.cu:138-141 */
static inline float go_between(float u, float lo, float hi) { return u * (hi - lo) + lo; }

/* ------------------------------------------------------------------------------------------
 * M2: exp for the adaptive support weight (weight_cu, gipuma.cu:186-193).  x <= 0 in use.
 * ---------------------------------------------------------------------------------------- */
static inline float go_exp(float x)
{
    if (!(x >= -86.0f)) return 0.0f; /* also NaN */
    if (x > 86.0f) x = 86.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float e = fmaf(p, r * r, r) + 1.0f;
    union { float f; int32_t i; } b;
    b.f = e;
    b.i += ((int32_t)n) << 23;
    return b.f;
}

float gipuma_oracle_exp(float x) { return go_exp(x); }

/* ------------------------------------------------------------------------------------------
 * small vector / matrix helpers, literal operation order of config.h
 * ---------------------------------------------------------------------------------------- */
/* matvecmul4, config.h:163-176:  (m0*x + m1*y) + m2*z */
static inline void go_matvec(const float *m, const float v[3], float out[3])
{
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    out[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    out[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}

/* matmul_cu, config.h:205-241 */
static inline void go_matmul(const float *a, const float *b, float *o)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[c + 3] + a[3 * r + 2] * b[c + 6];
}

/* dot4, config.h:36-38 */
static inline float go_dot(const float a[3], const float b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* normalize_cu, gipuma.cu:113-120 (rsqrtf -> M2) */
static inline void go_normalize(float v[3])
{
    const float ns = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float inv = 1.0f / sqrtf(ns);
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

/* getViewVector_cu + get3Dpoint_cu1, gipuma.cu:80-89, 122-130 */
static inline void go_view_vector(const gipuma_hip_camera *cam, int x, int y, float v[3])
{
    float pt[3];
    pt[0] = (float)x - cam->P_col34[0];
    pt[1] = (float)y - cam->P_col34[1];
    pt[2] = 1.0f - cam->P_col34[2];
    go_matvec(cam->M_inv, pt, v);
    v[0] = v[0] - cam->C[0];
    v[1] = v[1] - cam->C[1];
    v[2] = v[2] - cam->C[2];
    go_normalize(v);
}

void gipuma_oracle_view_vector(const gipuma_hip_camera *cam, int x, int y, float v[3])
{
    go_view_vector(cam, x, y, v);
}

/* vecOnHemisphere_cu, gipuma.cu:131-137 */
static inline void go_on_hemisphere(float v[3], const float view[3])
{
    const float dp = go_dot(v, view);
    if (dp > 0.0f) {
        v[0] = -v[0];
        v[1] = -v[1];
        v[2] = -v[2];
    }
}

/* getD_cu, gipuma.cu:96-111 */
static inline float go_plane_d(const gipuma_hip_camera *cam, const float n[3], int x, int y,
                               float depth)
{
    float pt[3], X[3];
    pt[0] = depth * (float)x - cam->P_col34[0];
    pt[1] = depth * (float)y - cam->P_col34[1];
    pt[2] = depth - cam->P_col34[2];
    go_matvec(cam->M_inv, pt, X);
    return -(go_dot(n, X));
}

float gipuma_oracle_plane_d(const gipuma_hip_camera *cam, const float n[3], int x, int y,
                            float depth)
{
    return go_plane_d(cam, n, x, y, depth);
}

/* getDisparity_cu / getDepthFromPlane3_cu, gipuma.cu:694-715 */
static inline float go_depth_from_plane(const gipuma_hip_camera *cam, const float pl[4], int x,
                                        int y)
{
    const float d = pl[3];
    if (d != d) return 1000.0f;
    return -d * cam->fx /
           ((pl[0] * ((float)x - cam->K[2])) + (pl[1] * ((float)y - cam->K[5])) * cam->alpha +
            pl[2] * cam->fx);
}

float gipuma_oracle_depth_from_plane(const gipuma_hip_camera *cam, const float plane[4], int x,
                                     int y)
{
    return go_depth_from_plane(cam, plane, x, y);
}

/* disparityDepthConversion_cu, gipuma.cu:66-68 */
static inline float go_disp_depth(float f, float baseline, float d) { return f * baseline / d; }

/* getHomography_cu, gipuma.cu:339-356:  H = K_to * ((R_to - t_to n^T / d) * K_ref^-1) */
static inline void go_homography(const gipuma_hip_camera *ref, const gipuma_hip_camera *to,
                                 const float n[3], float dpl, float H[9])
{
    float tmp[9], tmp2[9];
    /* outer_product4, config.h:85-94 */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) tmp[3 * r + c] = to->t[r] * n[c];
    /* matdivide, config.h:139-148 */
    for (int k = 0; k < 9; k++) tmp[k] = tmp[k] / dpl;
    /* matmatsub2, config.h:127-136 */
    for (int k = 0; k < 9; k++) tmp[k] = to->R[k] - tmp[k];
    go_matmul(tmp, ref->K_inv, tmp2);
    go_matmul(to->K, tmp2, H);
}

void gipuma_oracle_homography(const gipuma_hip_camera *ref, const gipuma_hip_camera *to,
                              const float n[3], float dpl, float H[9])
{
    go_homography(ref, to, n, dpl, H);
}

/* ------------------------------------------------------------------------------------------
 * M1: image access
 * ---------------------------------------------------------------------------------------- */
static inline int go_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* point sample with clamp-to-edge: tex2D(tex, i+0.5, j+0.5) of config.h:245-248 for integer
 * (i,j); also what the shared-memory tile holds (gipuma.cu:1393-1402) */
static inline float go_texel_s(const float *img, int rows, int cols, int pitch, int stride, int x, int y)
{
    x = go_clampi(x, 0, cols - 1);
    y = go_clampi(y, 0, rows - 1);
    return img[(size_t)y * (size_t)pitch + (size_t)x * (size_t)stride];
}
static inline float go_texel(const float *img, int rows, int cols, int pitch, int x, int y)
{
    return go_texel_s(img, rows, cols, pitch, 1, x, y);
}

static inline float go_lerp(float a, float t0, float t1) { return fmaf(a, t1 - t0, t0); }

/* The five bilinear taps of pmCostComputation_shared (gipuma.cu:251-253) at source position
 * (x,y): out = { centre, x+1, x-1, y+1, y-1 }.  Model M1. */
static inline void go_sample5_s(const float *img, int rows, int cols, int pitch, int stride, float x,
                                float y, float out[5])
{
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float a = x - fx0, b = y - fy0;
    /* keep the float->int conversion defined for huge / non-finite coordinates */
    const int ix = (int)fminf(fmaxf(fx0, -2.0f), (float)cols);
    const int iy = (int)fminf(fmaxf(fy0, -2.0f), (float)rows);
    float t[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t[r][c] = go_texel_s(img, rows, cols, pitch, stride, ix - 1 + c, iy - 1 + r);
    const float C0 = go_lerp(a, t[0][1], t[0][2]);
    const float L1 = go_lerp(a, t[1][0], t[1][1]);
    const float C1 = go_lerp(a, t[1][1], t[1][2]);
    const float R1 = go_lerp(a, t[1][2], t[1][3]);
    const float L2 = go_lerp(a, t[2][0], t[2][1]);
    const float C2 = go_lerp(a, t[2][1], t[2][2]);
    const float R2 = go_lerp(a, t[2][2], t[2][3]);
    const float C3 = go_lerp(a, t[3][1], t[3][2]);
    out[0] = go_lerp(b, C1, C2);
    out[1] = go_lerp(b, R1, R2);
    out[2] = go_lerp(b, L1, L2);
    out[3] = go_lerp(b, C2, C3);
    out[4] = go_lerp(b, C0, C1);
}

static inline void go_sample5(const float *img, int rows, int cols, int pitch, float x, float y,
                              float out[5])
{
    go_sample5_s(img, rows, cols, pitch, 1, x, y, out);
}

void gipuma_oracle_sample5(const float *img, int rows, int cols, int pitch, float x, float y,
                           float out[5])
{
    go_sample5(img, rows, cols, pitch, x, y, out);
}

/* The taps as the cost functions use them (model M1, the kernels' taps12 in pm_core.h): out = { centre value,
 * I(x+1, y) - I(x-1, y), I(x, y+1) - I(x, y-1) } of the bilinear image at (x, y), from the same 4x4 texel window as
 * go_sample5_s (t[r][c]: row iy-1+r, column ix-1+c; the corners are not read), columns interpolated along y first and the
 * two differences taken on the texels:
 *     V_c = lerp(b, t[1][c], t[2][c])  c = 0..3        W_c = lerp(b, t[2][c] - t[0][c], t[3][c] - t[1][c])  c = 1, 2
 *     centre = lerp(a, V_1, V_2)     d/dx = lerp(a, V_2 - V_0, V_3 - V_1)     d/dy = lerp(a, W_1, W_2)
 * In exact arithmetic these are out[0], out[1] - out[2] and out[3] - out[4] of go_sample5_s. */
static inline void go_taps3_s(const float *img, int rows, int cols, int pitch, int stride, float x, float y,
                              float out[3])
{
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float a = x - fx0, b = y - fy0;
    const int ix = (int)fminf(fmaxf(fx0, -2.0f), (float)cols);
    const int iy = (int)fminf(fmaxf(fy0, -2.0f), (float)rows);
    float t[4][4];
    if (ix >= 1 && ix <= cols - 3 && iy >= 1 && iy <= rows - 3) { /* (the whole window inside the image: no clamping; same texels) */
        const float *p = img + (size_t)(iy - 1) * (size_t)pitch + (size_t)(ix - 1) * (size_t)stride;
        for (int r = 0; r < 4; r++, p += pitch)
            for (int c = 0; c < 4; c++) t[r][c] = p[c * stride];
    } else {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) t[r][c] = go_texel_s(img, rows, cols, pitch, stride, ix - 1 + c, iy - 1 + r);
    }
    const float V0 = go_lerp(b, t[1][0], t[2][0]), V1 = go_lerp(b, t[1][1], t[2][1]);
    const float V2 = go_lerp(b, t[1][2], t[2][2]), V3 = go_lerp(b, t[1][3], t[2][3]);
    const float W1 = go_lerp(b, t[2][1] - t[0][1], t[3][1] - t[1][1]);
    const float W2 = go_lerp(b, t[2][2] - t[0][2], t[3][2] - t[1][2]);
    out[0] = go_lerp(a, V1, V2);
    out[1] = go_lerp(a, V2 - V0, V3 - V1);
    out[2] = go_lerp(a, W1, W2);
}

void gipuma_oracle_taps3(const float *img, int rows, int cols, int pitch, float x, float y, float out[3])
{
    go_taps3_s(img, rows, cols, pitch, 1, x, y, out);
}

/* ------------------------------------------------------------------------------------------
 * FLAVOURS of the per-sample arithmetic.
 * The reference's source leaves three things to nvcc and the texture unit; each has a "model" form (rounds 1-5: M1-M3 above)
 * and the LITERAL form -- the operation order of the source as g++ compiles it in oracle/_ref (no contraction, IEEE division,
 * one tex2D call per tap):
 *   GO_LIT_TAPS (1)  every one of the five taps of gipuma.cu:251-253 is its own bilinear fetch at the coordinates the source
 *                    writes, (pt.x +- 1 + 0.5f, pt.y + 0.5f), with xB = x - 0.5f, i = floor(xB), a = xB - i recomputed per tap:
 *                    the +-1 taps' fractions differ from the centre's in the last bits (two roundings each)  [else M1];
 *   GO_LIT_DIV  (2)  x / z and y / z as IEEE divisions (vecdiv4, config.h:44-47)  [else M2: x * (1/z)];
 *   GO_LIT_FMA  (4)  H*(x,y,1) as m0*x + m1*y + m2 (matvecmul4noz, config.h:150-162), dis = (1-alpha)*colDis + alpha*gradDis
 *                    (gipuma.cu:272), cost = cost + w*dis (:274, :672), all unfused  [else M3: the fmaf nesting].
 * Flavour 7 is the reference's own source order: it reproduces oracle/_ref in every bit (tests/test_oracle_vs_ref.py).
 * Flavour 6 -- division and multiply-adds literal, the model's taps -- is the DEFAULT since round 6: it is what the kernels'
 * default mode computes (pm_sample.h, PM_MODEL 6); flavour 0 is the model of rounds 1-5, what GIPUMA_HIP_FLAG_FAST's kernels
 * approximate.  gipuma_oracle_set_flavour(mask) selects; gray and colour alike.  The bilinear lerp itself
 * (fmaf(a, t1 - t0, t0), exact fp32 weights) is the shim's stand-in for the texture unit in all of them.
 * ---------------------------------------------------------------------------------------- */
#define GO_LIT_TAPS 1
#define GO_LIT_DIV 2
#define GO_LIT_FMA 4
#define GO_DEFAULT_FLAVOUR (GO_LIT_DIV | GO_LIT_FMA)
static int go_flavour = GO_DEFAULT_FLAVOUR;
void gipuma_oracle_set_flavour(int mask) { go_flavour = mask < 0 ? GO_DEFAULT_FLAVOUR : (mask & 7); }
int gipuma_oracle_get_flavour(void) { return go_flavour; }
int gipuma_oracle_default_flavour(void) { return GO_DEFAULT_FLAVOUR; }

/* one bilinear fetch tex2D(r, x, y) of the shim's texture model (ref_harness.cpp: ref_tex_channel); `stride` floats per texel */
static inline float go_tex2d_s(const float *img, int rows, int cols, int pitch, int stride, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = xb - fx, b = yb - fy;
    const int ix = (int)fminf(fmaxf(fx, -2.0f), (float)cols);
    const int iy = (int)fminf(fmaxf(fy, -2.0f), (float)rows);
    const float t00 = go_texel_s(img, rows, cols, pitch, stride, ix, iy), t10 = go_texel_s(img, rows, cols, pitch, stride, ix + 1, iy);
    const float t01 = go_texel_s(img, rows, cols, pitch, stride, ix, iy + 1), t11 = go_texel_s(img, rows, cols, pitch, stride, ix + 1, iy + 1);
    const float r0 = fmaf(a, t10 - t00, t00), r1 = fmaf(a, t11 - t01, t01);
    return fmaf(b, r1 - r0, r0);
}

/* getCorrespondingPoint_cu, gipuma.cu:207-217: the warp of q = (qx, qy) through H */
static inline void go_warp(const float *H, float qx, float qy, int fl, float *sx, float *sy)
{
    float X, Y, Z;
    if (fl & GO_LIT_FMA) { /* matvecmul4noz, config.h:150-162 */
        X = H[0] * qx + H[1] * qy + H[2];
        Y = H[3] * qx + H[4] * qy + H[5];
        Z = H[6] * qx + H[7] * qy + H[8];
    } else { /* M3 */
        X = fmaf(H[1], qy, fmaf(H[0], qx, H[2]));
        Y = fmaf(H[4], qy, fmaf(H[3], qx, H[5]));
        Z = fmaf(H[7], qy, fmaf(H[6], qx, H[8]));
    }
    if (fl & GO_LIT_DIV) { /* vecdiv4, config.h:44-47 */
        *sx = X / Z;
        *sy = Y / Z;
    } else { /* M2 */
        const float rz = 1.0f / Z;
        *sx = X * rz;
        *sy = Y * rz;
    }
}

/* the source-side terms of pmCostComputation_shared at (sx, sy): out = { I2(x, y), I2(x+1, y) - I2(x-1, y), I2(x, y+1) - I2(x, y-1) } */
static inline void go_source_terms(const float *img, int rows, int cols, int pitch, int stride, float sx, float sy, int fl,
                                   float out[3])
{
    if (fl & GO_LIT_TAPS) { /* gipuma.cu:251-253, argument expressions as written */
        out[1] = go_tex2d_s(img, rows, cols, pitch, stride, sx + 1 + 0.5f, sy + 0.5f) -
                 go_tex2d_s(img, rows, cols, pitch, stride, sx - 1 + 0.5f, sy + 0.5f);
        out[2] = go_tex2d_s(img, rows, cols, pitch, stride, sx + 0.5f, sy + 1 + 0.5f) -
                 go_tex2d_s(img, rows, cols, pitch, stride, sx + 0.5f, sy - 1 + 0.5f);
        out[0] = go_tex2d_s(img, rows, cols, pitch, stride, sx + 0.5f, sy + 0.5f);
    } else { /* M1 */
        go_taps3_s(img, rows, cols, pitch, stride, sx, sy, out);
    }
}

/* dis = (1 - alpha) colDis + alpha gradDis (gipuma.cu:272) and cost + w dis (:274, :672) */
static inline float go_dis_mix(float alpha, float colDis, float gradDis, int fl)
{
    if (fl & GO_LIT_FMA) return (1.f - alpha) * colDis + alpha * gradDis;
    return fmaf(alpha, gradDis, (1.f - alpha) * colDis);
}
static inline float go_accum(float w, float dis, float cost, int fl)
{
    if (fl & GO_LIT_FMA) return cost + w * dis;
    return fmaf(w, dis, cost);
}

/* ------------------------------------------------------------------------------------------
 * patch cost of one view: pmCost_shared + pmCostComputation_shared, gipuma.cu:585-680, 223-277
 * (pmCost/pmCostComputation, :455-518/:278-320, used by the init kernel, are the same function
 * of the images -- SURVEY 3.4)
 * ---------------------------------------------------------------------------------------- */
/* dis of one window sample with the reference-side terms handed in: the warp of q = (ix, iy) through H
 * (getCorrespondingPoint_cu, gipuma.cu:207-217), the five bilinear taps and the truncated colour / gradient differences of
 * pmCostComputation_shared (gipuma.cu:251-274).  A function of (q, H, view) only -- not of the pixel whose window q belongs
 * to (what pm_push.h builds on, see gipuma_oracle_push_costs). */
static inline float go_dis_cached(const gipuma_hip_desc *d, int view, const float *H, int ix, int iy, float leftValue,
                                  float gx1, float gy1, int fl)
{
    const gipuma_hip_params *ap = &d->params;
    float sx, sy, s[3];
    go_warp(H, (float)ix, (float)iy, fl, &sx, &sy);
    go_source_terms(d->images[view], d->rows, d->cols, d->pitch, 1, sx, sy, fl, s);
    const float colDiff = fabsf(leftValue - s[0]);
    const float gradX = gx1 - s[1];
    const float gradY = gy1 - s[2];
    const float gradDis = fminf((fabsf(gradX) + fabsf(gradY)) * 0.0625f, ap->tau_gradient);
    const float colDis = fminf(colDiff, ap->tau_color);
    return go_dis_mix(ap->alpha, colDis, gradDis, fl);
}
static inline float go_dis_at(const gipuma_hip_desc *d, int view, const float *H, int ix, int iy,
                              float leftValue)
{
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *ref = d->images[0];
    const float up = go_texel(ref, rows, cols, pitch, ix, iy - 1);
    const float down = go_texel(ref, rows, cols, pitch, ix, iy + 1);
    const float left = go_texel(ref, rows, cols, pitch, ix - 1, iy);
    const float right = go_texel(ref, rows, cols, pitch, ix + 1, iy);
    return go_dis_cached(d, view, H, ix, iy, leftValue, right - left, down - up, go_flavour);
}

/* What a pixel's window contributes to EVERY hypothesis and view evaluated at that pixel -- the support weights
 * (weight_cu, gipuma.cu:186-193) and the reference-side terms of pmCostComputation_shared (:254-259: the texel and its
 * central differences) -- computed once per pixel visit instead of once per (hypothesis, view): 11 hypotheses x N views
 * per half-sweep read them.  Same values, same operation order per sample: the results do not change by a bit; the oracle
 * runs about twice as fast.  One cache per thread; `go_epoch` (bumped by every exported entry point) keeps a cache from
 * outliving the images or parameters it was computed from. */
#define GO_MAXWIN (25 * 25) /* box <= 49 */
typedef struct {
    const gipuma_hip_desc *d;
    unsigned epoch;
    int px, py;
    float centre;
    float w[GO_MAXWIN + 8], I[GO_MAXWIN + 8], gx1[GO_MAXWIN + 8], gy1[GO_MAXWIN + 8]; /* (+8: the 8-wide path reads whole vectors) */
} go_pixel_cache;
static _Thread_local go_pixel_cache go_pc = {0, 0, -1, -1, 0.0f, {0}, {0}, {0}, {0}};
static unsigned go_epoch = 1;
#define GO_ENTER() (go_epoch++)

static const go_pixel_cache *go_cache_for(const gipuma_hip_desc *d, int px, int py)
{
    go_pixel_cache *pc = &go_pc;
    if (pc->d == d && pc->epoch == go_epoch && pc->px == px && pc->py == py) return pc;
    const gipuma_hip_params *ap = &d->params;
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *ref = d->images[0];
    const int hRad = (ap->box_hsize - 1) / 2, vRad = (ap->box_vsize - 1) / 2;
    const float gamma = ap->gamma;
    pc->d = d;
    pc->epoch = go_epoch;
    pc->px = px;
    pc->py = py;
    pc->centre = go_texel(ref, rows, cols, pitch, px, py);
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        for (int j = -vRad; j < vRad + 1; j += GO_WIN_INCREMENT, k++) {
            const int ix = px + i, iy = py + j;
            const float leftValue = go_texel(ref, rows, cols, pitch, ix, iy);
            const float colorDis = fabsf(leftValue - pc->centre);
            pc->I[k] = leftValue;
            pc->w[k] = go_exp(-colorDis / gamma);
            pc->gx1[k] = go_texel(ref, rows, cols, pitch, ix + 1, iy) - go_texel(ref, rows, cols, pitch, ix - 1, iy);
            pc->gy1[k] = go_texel(ref, rows, cols, pitch, ix, iy + 1) - go_texel(ref, rows, cols, pitch, ix, iy - 1);
        }
    }
    return pc;
}

/* ------------------------------------------------------------------------------------------
 * The same sample arithmetic eight window rows at a time (AVX2 + FMA: the flags this file is built with).  Every lane
 * executes the operations of go_warp / go_taps3_s / go_dis_cached in their order -- IEEE add, mul, div, fma, floor and
 * min/max with the scalar functions' NaN behaviour (vmaxps / vminps return their SECOND operand when one is NaN, as
 * fmaxf(x, c) / fminf(x, c) return c) -- so a lane's dis equals the scalar function's bit for bit; the chain over the
 * window (columns outer, rows inner, gipuma.cu:633-676) stays scalar and sequential.  Flavours with the literal taps
 * (bit 0) keep the scalar path.  gipuma_oracle_set_simd(0) switches back to the scalar functions (tests compare the two).
 * ---------------------------------------------------------------------------------------- */
static int go_simd = GO_HAVE_AVX2;
void gipuma_oracle_set_simd(int on) { go_simd = on && GO_HAVE_AVX2; }
int gipuma_oracle_get_simd(void) { return go_simd; }

#if GO_HAVE_AVX2
typedef struct {
    __m256 sx, sy;
} go_v2;
static inline go_v2 go_warp8(const float *H, float qx, __m256 qy, int fl)
{
    __m256 X, Y, Z;
    if (fl & GO_LIT_FMA) {
        X = _mm256_add_ps(_mm256_add_ps(_mm256_set1_ps(H[0] * qx), _mm256_mul_ps(_mm256_set1_ps(H[1]), qy)), _mm256_set1_ps(H[2]));
        Y = _mm256_add_ps(_mm256_add_ps(_mm256_set1_ps(H[3] * qx), _mm256_mul_ps(_mm256_set1_ps(H[4]), qy)), _mm256_set1_ps(H[5]));
        Z = _mm256_add_ps(_mm256_add_ps(_mm256_set1_ps(H[6] * qx), _mm256_mul_ps(_mm256_set1_ps(H[7]), qy)), _mm256_set1_ps(H[8]));
    } else {
        X = _mm256_fmadd_ps(_mm256_set1_ps(H[1]), qy, _mm256_set1_ps(fmaf(H[0], qx, H[2])));
        Y = _mm256_fmadd_ps(_mm256_set1_ps(H[4]), qy, _mm256_set1_ps(fmaf(H[3], qx, H[5])));
        Z = _mm256_fmadd_ps(_mm256_set1_ps(H[7]), qy, _mm256_set1_ps(fmaf(H[6], qx, H[8])));
    }
    go_v2 o;
    if (fl & GO_LIT_DIV) {
        o.sx = _mm256_div_ps(X, Z);
        o.sy = _mm256_div_ps(Y, Z);
    } else {
        const __m256 rz = _mm256_div_ps(_mm256_set1_ps(1.0f), Z);
        o.sx = _mm256_mul_ps(X, rz);
        o.sy = _mm256_mul_ps(Y, rz);
    }
    return o;
}
static inline __m256 go_lerp8(__m256 a, __m256 t0, __m256 t1) { return _mm256_fmadd_ps(a, _mm256_sub_ps(t1, t0), t0); }
static inline __m256 go_abs8(__m256 x) { return _mm256_andnot_ps(_mm256_set1_ps(-0.0f), x); }
/* go_taps3_s for eight positions: out = { centre, d/dx, d/dy } (model M1) */
static inline void go_taps3_8(const float *img, int rows, int cols, int pitch, int stride, __m256 x, __m256 y, __m256 out[3])
{
    const __m256 fx0 = _mm256_floor_ps(x), fy0 = _mm256_floor_ps(y);
    const __m256 a = _mm256_sub_ps(x, fx0), b = _mm256_sub_ps(y, fy0);
    const __m256i ix = _mm256_cvttps_epi32(_mm256_min_ps(_mm256_max_ps(fx0, _mm256_set1_ps(-2.0f)), _mm256_set1_ps((float)cols)));
    const __m256i iy = _mm256_cvttps_epi32(_mm256_min_ps(_mm256_max_ps(fy0, _mm256_set1_ps(-2.0f)), _mm256_set1_ps((float)rows)));
    const __m256i zero = _mm256_setzero_si256(), cmax = _mm256_set1_epi32(cols - 1), rmax = _mm256_set1_epi32(rows - 1);
    __m256i cx[4], ry[4];
    for (int c = 0; c < 4; c++) {
        const __m256i xc = _mm256_min_epi32(_mm256_max_epi32(_mm256_add_epi32(ix, _mm256_set1_epi32(c - 1)), zero), cmax);
        cx[c] = _mm256_mullo_epi32(xc, _mm256_set1_epi32(stride));
        const __m256i yc = _mm256_min_epi32(_mm256_max_epi32(_mm256_add_epi32(iy, _mm256_set1_epi32(c - 1)), zero), rmax);
        ry[c] = _mm256_mullo_epi32(yc, _mm256_set1_epi32(pitch));
    }
#define GO_T(r, c) _mm256_i32gather_ps(img, _mm256_add_epi32(ry[r], cx[c]), 4)
    const __m256 t01 = GO_T(0, 1), t02 = GO_T(0, 2);
    const __m256 t10 = GO_T(1, 0), t11 = GO_T(1, 1), t12 = GO_T(1, 2), t13 = GO_T(1, 3);
    const __m256 t20 = GO_T(2, 0), t21 = GO_T(2, 1), t22 = GO_T(2, 2), t23 = GO_T(2, 3);
    const __m256 t31 = GO_T(3, 1), t32 = GO_T(3, 2);
#undef GO_T
    const __m256 V0 = go_lerp8(b, t10, t20), V1 = go_lerp8(b, t11, t21), V2 = go_lerp8(b, t12, t22), V3 = go_lerp8(b, t13, t23);
    const __m256 W1 = go_lerp8(b, _mm256_sub_ps(t21, t01), _mm256_sub_ps(t31, t11));
    const __m256 W2 = go_lerp8(b, _mm256_sub_ps(t22, t02), _mm256_sub_ps(t32, t12));
    out[0] = go_lerp8(a, V1, V2);
    out[1] = go_lerp8(a, _mm256_sub_ps(V2, V0), _mm256_sub_ps(V3, V1));
    out[2] = go_lerp8(a, W1, W2);
}
static inline __m256 go_dis_mix8(float alpha, __m256 colDis, __m256 gradDis, int fl)
{
    const __m256 oma = _mm256_set1_ps(1.f - alpha), al = _mm256_set1_ps(alpha);
    if (fl & GO_LIT_FMA) return _mm256_add_ps(_mm256_mul_ps(oma, colDis), _mm256_mul_ps(al, gradDis));
    return _mm256_fmadd_ps(al, gradDis, _mm256_mul_ps(oma, colDis));
}
/* the eight row coordinates (float)(py + j), j = j0, j0 + 2, ..., exact small integers */
static inline __m256 go_rows8(int py, int j0)
{
    return _mm256_cvtepi32_ps(_mm256_add_epi32(_mm256_set1_epi32(py + j0), _mm256_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14)));
}

static float go_view_cost_simd(const gipuma_hip_desc *d, int view, int px, int py, const float *H, const go_pixel_cache *pc, int fl)
{
    const gipuma_hip_params *ap = &d->params;
    const int hRad = (ap->box_hsize - 1) / 2, vRad = (ap->box_vsize - 1) / 2;
    const int nj = vRad + 1; /* rows of the window: j = -vRad, -vRad + 2, ..., vRad */
    const __m256 tau_g = _mm256_set1_ps(ap->tau_gradient), tau_c = _mm256_set1_ps(ap->tau_color);
    float cost = 0.0f;
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        const float qx = (float)(px + i);
        for (int jb = 0; jb < nj; jb += 8) {
            const go_v2 s = go_warp8(H, qx, go_rows8(py, -vRad + 2 * jb), fl);
            __m256 t[3];
            go_taps3_8(d->images[view], d->rows, d->cols, d->pitch, 1, s.sx, s.sy, t);
            const __m256 colDiff = go_abs8(_mm256_sub_ps(_mm256_loadu_ps(pc->I + k), t[0]));
            const __m256 gradX = _mm256_sub_ps(_mm256_loadu_ps(pc->gx1 + k), t[1]);
            const __m256 gradY = _mm256_sub_ps(_mm256_loadu_ps(pc->gy1 + k), t[2]);
            const __m256 gradDis = _mm256_min_ps(_mm256_mul_ps(_mm256_add_ps(go_abs8(gradX), go_abs8(gradY)), _mm256_set1_ps(0.0625f)), tau_g);
            const __m256 colDis = _mm256_min_ps(colDiff, tau_c);
            float dis[8];
            _mm256_storeu_ps(dis, go_dis_mix8(ap->alpha, colDis, gradDis, fl));
            const int n = nj - jb < 8 ? nj - jb : 8;
            for (int l = 0; l < n; l++, k++) cost = go_accum(pc->w[k], dis[l], cost, fl);
        }
    }
    return cost;
}
#endif

static float go_view_cost(const gipuma_hip_desc *d, int view, int px, int py, const float pl[4])
{
    const gipuma_hip_params *ap = &d->params;
    const int hRad = (ap->box_hsize - 1) / 2; /* gipuma.cu:1474 (init uses box/2, same for odd) */
    const int vRad = (ap->box_vsize - 1) / 2;
    const int fl = go_flavour;
    if ((hRad + 1) * (vRad + 1) > GO_MAXWIN) return GO_MAXCOST; /* (box <= 49, like the library: go_check) */

    float H[9];
    go_homography(&d->cameras[0], &d->cameras[view], pl, pl[3], H);

    const go_pixel_cache *pc = go_cache_for(d, px, py);
#if GO_HAVE_AVX2
    if (go_simd && !(fl & GO_LIT_TAPS)) return go_view_cost_simd(d, view, px, py, H, pc, fl);
#endif
    float cost = 0.0f;
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        for (int j = -vRad; j < vRad + 1; j += GO_WIN_INCREMENT, k++) {
            /* weight_cu, gipuma.cu:186-193: pc->w[k]; pmCostComputation_shared, :251-274 */
            const float dis = go_dis_cached(d, view, H, px + i, py + j, pc->I[k], pc->gx1[k], pc->gy1[k], fl);
            cost = go_accum(pc->w[k], dis, cost, fl);
        }
    }
    return cost;
}

/* l1_norm(float4), gipuma.cu:174-179: mean |.| of x,y,z; the float4 operators zero .w
 * (vector_operations.h:9-14) */
static inline float go_l1_3(const float v[3])
{
    return (fabsf(v[0]) + fabsf(v[1]) + fabsf(v[2])) * 0.3333333f;
}

/* The same functions instantiated with T = float4 (-color_processing, gipuma.cu:1965-1968):
 * images are float4 per pixel (B, G, R, unset alpha: main.cpp:943-956), every image
 * difference is per channel and reduced by l1_norm(float4). */
/* the pixel cache for T = float4: per window sample the texel's three channels, their central differences (right - left,
 * down - up: the reference-side terms of pmCostComputation_shared<float4>) and the support weight */
typedef struct {
    const gipuma_hip_desc *d;
    unsigned epoch;
    int px, py;
    float w[GO_MAXWIN + 8], lv[GO_MAXWIN + 8][3], gx1[GO_MAXWIN + 8][3], gy1[GO_MAXWIN + 8][3]; /* (+8: whole vectors) */
} go_pixel_cache_c4;
static _Thread_local go_pixel_cache_c4 go_pc4 = {0, 0, -1, -1, {0}, {{0}}, {{0}}, {{0}}};

static const go_pixel_cache_c4 *go_cache_c4_for(const gipuma_hip_desc *d, int px, int py)
{
    go_pixel_cache_c4 *pc = &go_pc4;
    if (pc->d == d && pc->epoch == go_epoch && pc->px == px && pc->py == py) return pc;
    const gipuma_hip_params *ap = &d->params;
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *ref = d->images[0];
    const int hRad = (ap->box_hsize - 1) / 2, vRad = (ap->box_vsize - 1) / 2;
    const float gamma = ap->gamma;
    pc->d = d;
    pc->epoch = go_epoch;
    pc->px = px;
    pc->py = py;
    float centre[3];
    for (int c = 0; c < 3; c++) centre[c] = go_texel_s(ref + c, rows, cols, pitch, 4, px, py);
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        for (int j = -vRad; j < vRad + 1; j += GO_WIN_INCREMENT, k++) {
            const int ix = px + i, iy = py + j;
            float dc[3];
            for (int c = 0; c < 3; c++) {
                pc->lv[k][c] = go_texel_s(ref + c, rows, cols, pitch, 4, ix, iy);
                dc[c] = pc->lv[k][c] - centre[c];
                pc->gx1[k][c] = go_texel_s(ref + c, rows, cols, pitch, 4, ix + 1, iy) - go_texel_s(ref + c, rows, cols, pitch, 4, ix - 1, iy);
                pc->gy1[k][c] = go_texel_s(ref + c, rows, cols, pitch, 4, ix, iy + 1) - go_texel_s(ref + c, rows, cols, pitch, 4, ix, iy - 1);
            }
            pc->w[k] = go_exp(-go_l1_3(dc) / gamma);
        }
    }
    return pc;
}

#if GO_HAVE_AVX2
static inline __m256 go_l1_3_8(__m256 x, __m256 y, __m256 z)
{
    return _mm256_mul_ps(_mm256_add_ps(_mm256_add_ps(go_abs8(x), go_abs8(y)), go_abs8(z)), _mm256_set1_ps(0.3333333f));
}
static float go_view_cost_c4_simd(const gipuma_hip_desc *d, int view, int px, int py, const float *H, const go_pixel_cache_c4 *pc, int fl)
{
    const gipuma_hip_params *ap = &d->params;
    const float *src = d->images[view];
    const int hRad = (ap->box_hsize - 1) / 2, vRad = (ap->box_vsize - 1) / 2;
    const int nj = vRad + 1;
    const __m256 tau_g = _mm256_set1_ps(ap->tau_gradient), tau_c = _mm256_set1_ps(ap->tau_color);
    const __m256i k3 = _mm256_setr_epi32(0, 3, 6, 9, 12, 15, 18, 21);
    float cost = 0.0f;
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        const float qx = (float)(px + i);
        for (int jb = 0; jb < nj; jb += 8) {
            const go_v2 s = go_warp8(H, qx, go_rows8(py, -vRad + 2 * jb), fl);
            __m256 cd[3], gX[3], gY[3];
            for (int c = 0; c < 3; c++) {
                __m256 t[3];
                go_taps3_8(src + c, d->rows, d->cols, d->pitch, 4, s.sx, s.sy, t);
                cd[c] = _mm256_sub_ps(_mm256_i32gather_ps(&pc->lv[k][c], k3, 4), t[0]);
                gX[c] = _mm256_sub_ps(_mm256_i32gather_ps(&pc->gx1[k][c], k3, 4), t[1]);
                gY[c] = _mm256_sub_ps(_mm256_i32gather_ps(&pc->gy1[k][c], k3, 4), t[2]);
            }
            const __m256 colDiff = go_l1_3_8(cd[0], cd[1], cd[2]);
            const __m256 gradDis = _mm256_min_ps(_mm256_mul_ps(_mm256_add_ps(go_l1_3_8(gX[0], gX[1], gX[2]), go_l1_3_8(gY[0], gY[1], gY[2])),
                                                               _mm256_set1_ps(0.0625f)), tau_g);
            const __m256 colDis = _mm256_min_ps(colDiff, tau_c);
            float dis[8];
            _mm256_storeu_ps(dis, go_dis_mix8(ap->alpha, colDis, gradDis, fl));
            const int n = nj - jb < 8 ? nj - jb : 8;
            for (int l = 0; l < n; l++, k++) cost = go_accum(pc->w[k], dis[l], cost, fl);
        }
    }
    return cost;
}
#endif

static float go_view_cost_c4(const gipuma_hip_desc *d, int view, int px, int py, const float pl[4])
{
    const gipuma_hip_params *ap = &d->params;
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *src = d->images[view];
    const int hRad = (ap->box_hsize - 1) / 2;
    const int vRad = (ap->box_vsize - 1) / 2;
    const float alpha = ap->alpha, tau_color = ap->tau_color, tau_gradient = ap->tau_gradient;
    if ((hRad + 1) * (vRad + 1) > GO_MAXWIN) return GO_MAXCOST;

    float H[9];
    go_homography(&d->cameras[0], &d->cameras[view], pl, pl[3], H);

    const go_pixel_cache_c4 *pc = go_cache_c4_for(d, px, py);
    const int fl = go_flavour;
#if GO_HAVE_AVX2
    if (go_simd && !(fl & GO_LIT_TAPS)) return go_view_cost_c4_simd(d, view, px, py, H, pc, fl);
#endif
    float cost = 0.0f;
    int k = 0;
    for (int i = -hRad; i < hRad + 1; i += GO_WIN_INCREMENT) {
        for (int j = -vRad; j < vRad + 1; j += GO_WIN_INCREMENT, k++) {
            float sx, sy;
            go_warp(H, (float)(px + i), (float)(py + j), fl, &sx, &sy);
            float cd[3], gradX[3], gradY[3];
            for (int c = 0; c < 3; c++) {
                float s[3];
                go_source_terms(src + c, rows, cols, pitch, 4, sx, sy, fl, s);
                cd[c] = pc->lv[k][c] - s[0];
                gradX[c] = pc->gx1[k][c] - s[1];
                gradY[c] = pc->gy1[k][c] - s[2];
            }
            const float colDiff = go_l1_3(cd);
            const float gradDis = fminf((go_l1_3(gradX) + go_l1_3(gradY)) * 0.0625f, tau_gradient);
            const float colDis = fminf(colDiff, tau_color);
            cost = go_accum(pc->w[k], go_dis_mix(alpha, colDis, gradDis, fl), cost, fl);
        }
    }
    return cost;
}

static float go_view_cost_any(const gipuma_hip_desc *d, int view, int px, int py, const float pl[4])
{
    if (d->channels == 4) return go_view_cost_c4(d, view, px, py, pl);
    return go_view_cost(d, view, px, py, pl);
}

float gipuma_oracle_view_cost(const gipuma_hip_desc *d, int view, int x, int y,
                              const float plane[4])
{
    GO_ENTER();
    return go_view_cost_any(d, view, x, y, plane);
}

/* sort_small, gipuma.cu:684-693 */
static void go_sort_small(float *v, int n)
{
    for (int i = 1; i < n; i++) {
        const float tmp = v[i];
        int j;
        for (j = i; j >= 1 && tmp < v[j - 1]; j--) v[j] = v[j - 1];
        v[j] = tmp;
    }
}

/* the combination half of pmCostMultiview_cu, gipuma.cu:769-805 */
static float go_aggregate(const float *view_costs, int n, int cost_comb, int n_best,
                          float good_factor)
{
    float cv[GIPUMA_HIP_MAX_VIEWS];
    int numValid = 0;
    for (int i = 0; i < n; i++) {
        float c = view_costs[i];
        if (c < GO_MAXCOST)
            numValid++;
        else
            c = GO_MAXCOST;
        cv[i] = c;
    }
    go_sort_small(cv, n);
    int numBest = numValid;
    if (cost_comb == GIPUMA_COMB_BEST_N) numBest = numBest < n_best ? numBest : n_best;
    if (cost_comb == GIPUMA_COMB_GOOD) numBest = n;
    const float costThresh = (n > 0 ? cv[0] : 0.0f) * good_factor;
    int numConsidered = 0;
    float cost = 0.0f;
    for (int i = 0; i < numBest; i++) {
        numConsidered++;
        float c = cv[i];
        if (cost_comb == GIPUMA_COMB_GOOD) c = fminf(c, costThresh);
        cost = cost + c;
    }
    cost = cost / ((float)numConsidered);
    if (numConsidered < 1) cost = GO_MAXCOST;
    if (cost != cost || cost > GO_MAXCOST || cost < 0) cost = GO_MAXCOST;
    return cost;
}

float gipuma_oracle_aggregate(const float *view_costs, int n, int cost_comb, int n_best,
                              float good_factor)
{
    return go_aggregate(view_costs, n, cost_comb, n_best, good_factor);
}

/* pmCostMultiview_cu, gipuma.cu:720-806 */
static float go_multiview_cost(const gipuma_hip_desc *d, int px, int py, const float pl[4])
{
    float cv[GIPUMA_HIP_MAX_VIEWS];
    const int n = d->n_selected;
    for (int i = 0; i < n; i++) cv[i] = go_view_cost_any(d, d->selected[i], px, py, pl);
    return go_aggregate(cv, n, d->params.cost_comb, d->params.n_best, d->params.good_factor);
}

float gipuma_oracle_multiview_cost(const gipuma_hip_desc *d, int x, int y, const float plane[4])
{
    GO_ENTER();
    return go_multiview_cost(d, x, y, plane);
}

/* The push formulation of gipuma_amd/csrc/pm_push.h restated on the CPU (gray images, square window):
 * the plane of producer (nx, ny) is a propagation candidate of its eight neighbours of the other
 * checkerboard colour -- slot c of pm::neighbour: the pixel that sees the producer up, down, left, right
 * at distance 1 (c = 0..3) and 5 (c = 4..7; gipuma.cu:1437-1462, 1571-1582).  dis is evaluated ONCE per
 * view on the stencil those eight windows share -- (N+5) x N points per family instead of 4 x N x N --,
 * then every consumer runs the reference's chain over its own window (its own support weights, columns
 * outer, rows inner, gipuma.cu:633-676) on the stored values, and its view costs are combined as in
 * pmCostMultiview_cu.  out[c] must equal go_multiview_cost(consumer c, plane) bit for bit wherever the
 * consumer lies inside the image; valid[c] = 0 elsewhere.  Returns the number of dis evaluations per view. */
int gipuma_oracle_push_costs(const gipuma_hip_desc *d, int nx, int ny, const float plane[4], float out[8],
                             int valid[8])
{
    const gipuma_hip_params *ap = &d->params;
    const int rows = d->rows, cols = d->cols, pitch = d->pitch;
    const float *ref = d->images[0];
    if (d->channels != 1 || ap->box_hsize != ap->box_vsize) return -1;
    const int R = (ap->box_hsize - 1) / 2, N = (2 * R) / GO_WIN_INCREMENT + 1, FW = N + 5; /* offsets -R, -R+2, ..., R */
    if (N * FW > 512) return -1;
    float cv[8][GIPUMA_HIP_MAX_VIEWS];
    int cdx[8], cdy[8];
    for (int c = 0; c < 8; c++) {
        const int dist = c < 4 ? 1 : 5, k = c & 3;
        cdx[c] = k == 2 ? dist : k == 3 ? -dist : 0; /* consumer = producer - neighbour offset of slot k */
        cdy[c] = k == 0 ? dist : k == 1 ? -dist : 0;
        const int px = nx + cdx[c], py = ny + cdy[c];
        valid[c] = px >= 0 && px < cols && py >= 0 && py < rows;
    }
    for (int v = 0; v < d->n_selected; v++) {
        const int view = d->selected[v];
        float H[9];
        go_homography(&d->cameras[0], &d->cameras[view], plane, plane[3], H);
        /* vertical family (consumers above / below): offsets (2x - R, 2y - R - 5), x < N, y < N + 5;
         * horizontal family: offsets (2x - R - 5, 2y - R), x < N + 5, y < N */
        float disV[512], disH[512];
        for (int y = 0; y < FW; y++)
            for (int x = 0; x < N; x++) {
                const int ix = nx + 2 * x - R, iy = ny + 2 * y - R - 5;
                disV[y * N + x] = go_dis_at(d, view, H, ix, iy, go_texel(ref, rows, cols, pitch, ix, iy));
            }
        for (int y = 0; y < N; y++)
            for (int x = 0; x < FW; x++) {
                const int ix = nx + 2 * x - R - 5, iy = ny + 2 * y - R;
                disH[y * FW + x] = go_dis_at(d, view, H, ix, iy, go_texel(ref, rows, cols, pitch, ix, iy));
            }
        for (int c = 0; c < 8; c++) {
            const int px = nx + cdx[c], py = ny + cdy[c];
            const float centre = go_texel(ref, rows, cols, pitch, px, py);
            float cost = 0.0f;
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) {
                    const float leftValue = go_texel(ref, rows, cols, pitch, px + 2 * i - R, py + 2 * j - R);
                    const float w = go_exp(-fabsf(leftValue - centre) / ap->gamma);
                    const float dis = cdx[c] == 0 ? disV[((cdy[c] + 5) / 2 + j) * N + i]
                                                  : disH[j * FW + (cdx[c] + 5) / 2 + i];
                    cost = go_accum(w, dis, cost, go_flavour);
                }
            cv[c][v] = cost;
        }
    }
    for (int c = 0; c < 8; c++)
        out[c] = go_aggregate(cv[c], d->n_selected, ap->cost_comb, ap->n_best, ap->good_factor);
    return 2 * N * FW;
}


/* ------------------------------------------------------------------------------------------
 * per-pixel stages
 * ---------------------------------------------------------------------------------------- */
/* gipuma_init_cu2, gipuma.cu:996-1051 */
static void go_init_pixel(const gipuma_hip_desc *d, int x, int y, float *norm4, float *cost)
{
    const gipuma_hip_camera *cam = &d->cameras[0];
    const size_t center = (size_t)y * (size_t)d->cols + (size_t)x;
    uint32_t draw = 0;
    float view[3];
    go_view_vector(cam, x, y, view);
    /* curand_between(mind, maxd), gipuma.cu:1028 */
    float disp = go_between(go_uniform(d->seed, 0, (uint32_t)x, (uint32_t)y, draw++),
                            d->params.min_disparity, d->params.max_disparity);
    /* rndUnitVectorSphereMarsaglia_cu, gipuma.cu:148-164 */
    float rx = 1.0f, ry = 1.0f, sum = 2.0f;
    while (sum >= 1.0f) {
        rx = go_between(go_uniform(d->seed, 0, (uint32_t)x, (uint32_t)y, draw++), -1.0f, 1.0f);
        ry = go_between(go_uniform(d->seed, 0, (uint32_t)x, (uint32_t)y, draw++), -1.0f, 1.0f);
        sum = rx * rx + ry * ry;
    }
    const float sq = sqrtf(1.0f - sum);
    float pl[4];
    pl[0] = 2.0f * rx * sq;
    pl[1] = 2.0f * ry * sq;
    pl[2] = 1.0f - 2.0f * sum;
    go_on_hemisphere(pl, view);
    const float depth = go_disp_depth(cam->f, cam->baseline, disp);
    pl[3] = go_plane_d(cam, pl, x, y, depth);
    memcpy(norm4 + 4 * center, pl, sizeof pl);
    cost[center] = go_multiview_cost(d, x, y, pl);
}

typedef struct {
    float pl[4];
    float cost;
    float depth; /* disp_now of the reference kernels (it is a depth) */
} go_pixel_state;

/* spatialPropagation_cu, gipuma.cu:832-874 */
static void go_try_neighbour(const gipuma_hip_desc *d, int x, int y, const float *norm4,
                             size_t nb, go_pixel_state *st)
{
    const gipuma_hip_camera *cam = &d->cameras[0];
    float cand[4];
    memcpy(cand, norm4 + 4 * nb, sizeof cand);
    const float depth_before = go_depth_from_plane(cam, cand, x, y);
    const float cost_before = go_multiview_cost(d, x, y, cand);
    if (depth_before >= cam->depth_min && depth_before <= cam->depth_max) {
        if (cost_before < st->cost) {
            st->depth = depth_before;
            memcpy(st->pl, cand, sizeof cand);
            st->cost = cost_before;
        }
    }
}

/* the neighbour tests of gipuma_checkerboard_spatialPropClose_cu (:1560-1582, distance 1)
 * and ..._spatialPropFar_cu (:1437-1462, distance 5): up, down, left, right */
static void go_propagate(const gipuma_hip_desc *d, int x, int y, const float *norm4, int dist,
                         go_pixel_state *st)
{
    const int rows = d->rows, cols = d->cols;
    const size_t center = (size_t)y * (size_t)cols + (size_t)x;
    if (y > dist - 1) go_try_neighbour(d, x, y, norm4, center - (size_t)dist * (size_t)cols, st);
    if (y < rows - dist) go_try_neighbour(d, x, y, norm4, center + (size_t)dist * (size_t)cols, st);
    if (x > dist - 1) go_try_neighbour(d, x, y, norm4, center - (size_t)dist, st);
    if (x < cols - dist) go_try_neighbour(d, x, y, norm4, center + (size_t)dist, st);
}

/* planeRefinement_cu + getRndDispAndUnitVector_cu, gipuma.cu:928-994, 890-927 */
static void go_refine(const gipuma_hip_desc *d, int x, int y, uint32_t phase, go_pixel_state *st)
{
    const gipuma_hip_camera *cam = &d->cameras[0];
    const gipuma_hip_params *ap = &d->params;
    float view[3];
    go_view_vector(cam, x, y, view);
    float deltaN = 1.0f;
    uint32_t draw = 0;
    const float maxdisp = ap->max_disparity / 2.0f;
    for (float deltaZ = maxdisp; deltaZ >= 0.01f; deltaZ = deltaZ / 10.0f) {
        /* getRndDispAndUnitVector_cu */
        const float disp = go_disp_depth(cam->f, cam->baseline, st->depth);
        const float minDelta = -fminf(deltaZ, ap->min_disparity + disp); /* sic: '+', :909 */
        const float maxDelta = fminf(deltaZ, ap->max_disparity - disp);
        const float u0 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
        const float u1 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
        const float u2 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
        const float u3 = go_uniform(d->seed, phase, (uint32_t)x, (uint32_t)y, draw++);
        const float dz = go_between(u0, minDelta, maxDelta);
        float dispOut = fminf(fmaxf(disp + dz, ap->min_disparity), ap->max_disparity);
        const float depthOut = go_disp_depth(cam->f, cam->baseline, dispOut);
        float cand[4];
        cand[0] = st->pl[0] + go_between(u1, -deltaN, deltaN);
        cand[1] = st->pl[1] + go_between(u2, -deltaN, deltaN);
        cand[2] = st->pl[2] + go_between(u3, -deltaN, deltaN);
        go_normalize(cand);
        go_on_hemisphere(cand, view);
        /* planeRefinement_cu body */
        cand[3] = go_plane_d(cam, cand, x, y, depthOut);
        const float c = go_multiview_cost(d, x, y, cand);
        if (c < st->cost) {
            st->cost = c;
            st->depth = depthOut;
            memcpy(st->pl, cand, sizeof cand);
        }
        deltaN = deltaN / 4.0f;
    }
}

int gipuma_oracle_refine_schedule(float max_disparity, float *delta_z, float *delta_n, int cap)
{
    int k = 0;
    float deltaN = 1.0f;
    for (float deltaZ = max_disparity / 2.0f; deltaZ >= 0.01f; deltaZ = deltaZ / 10.0f) {
        if (k < cap) {
            delta_z[k] = deltaZ;
            delta_n[k] = deltaN;
        }
        k++;
        deltaN = deltaN / 4.0f;
    }
    return k;
}

/* one pixel of one colour kernel: read state, run the requested stages, write back
 * (gipuma.cu:1527-1530 read, :1585-1587 write) */
static void go_sweep_pixel(const gipuma_hip_desc *d, int x, int y, float *norm4, float *cost,
                           uint32_t phase, unsigned stages)
{
    const size_t center = (size_t)y * (size_t)d->cols + (size_t)x;
    go_pixel_state st;
    memcpy(st.pl, norm4 + 4 * center, sizeof st.pl);
    st.cost = cost[center];
    st.depth = go_depth_from_plane(&d->cameras[0], st.pl, x, y);
    if (stages & GIPUMA_STAGE_CLOSE) go_propagate(d, x, y, norm4, 1, &st);
    if (stages & GIPUMA_STAGE_FAR) go_propagate(d, x, y, norm4, 5, &st);
    if (stages & GIPUMA_STAGE_REFINE) {
        /* the refine kernel re-derives disp_now from the stored plane, gipuma.cu:1660 */
        st.depth = go_depth_from_plane(&d->cameras[0], st.pl, x, y);
        go_refine(d, x, y, phase, &st);
    }
    cost[center] = st.cost;
    memcpy(norm4 + 4 * center, st.pl, sizeof st.pl);
}

/* ------------------------------------------------------------------------------------------
 * whole-image passes
 * ---------------------------------------------------------------------------------------- */
static int go_check(const gipuma_hip_desc *d)
{
    if (!d || d->abi_version != GIPUMA_HIP_ABI_VERSION) return GIPUMA_HIP_ERR_ARG;
    if (d->rows < 1 || d->cols < 1 || (d->channels != 1 && d->channels != 4) ||
        d->pitch < d->cols * d->channels)
        return GIPUMA_HIP_ERR_ARG;
    if (d->n_images < 1 || !d->images || !d->cameras) return GIPUMA_HIP_ERR_ARG;
    if (d->n_selected < 0 || d->n_selected > GIPUMA_HIP_MAX_VIEWS) return GIPUMA_HIP_ERR_ARG;
    for (int i = 0; i < d->n_selected; i++)
        if (d->selected[i] < 0 || d->selected[i] >= d->n_images) return GIPUMA_HIP_ERR_ARG;
    if (d->params.box_hsize < 1 || d->params.box_vsize < 1 || !(d->params.box_hsize & 1) ||
        !(d->params.box_vsize & 1))
        return GIPUMA_HIP_ERR_ARG;
    if (d->params.box_hsize > 49 || d->params.box_vsize > 49) return GIPUMA_HIP_ERR_UNSUPPORTED; /* like gipuma_hip_create */
    GO_ENTER(); /* a new call may see new images or parameters: the per-thread pixel caches start over */
    return 0;
}

int gipuma_oracle_init_planes(const gipuma_hip_desc *d, float *norm4, float *cost)
{
    int rc = go_check(d);
    if (rc) return rc;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < d->rows; y++)
        for (int x = 0; x < d->cols; x++) go_init_pixel(d, x, y, norm4, cost);
    return 0;
}

int gipuma_oracle_sweep(const gipuma_hip_desc *d, float *norm4, float *cost, int iteration,
                        int colour, unsigned stages, int unfused)
{
    int rc = go_check(d);
    if (rc) return rc;
    const uint32_t phase = go_phase(iteration, colour);
    /* black <=> (x+y) even, gipuma.cu:1730-1734; every pixel of a colour only reads pixels of
     * the other colour (distance 1 and 5 are odd), so the pixel order is free */
    if (unfused) {
        static const unsigned order[3] = {GIPUMA_STAGE_CLOSE, GIPUMA_STAGE_FAR,
                                          GIPUMA_STAGE_REFINE};
        for (int k = 0; k < 3; k++) {
            if (!(stages & order[k])) continue;
#pragma omp parallel for schedule(dynamic, 4)
            for (int y = 0; y < d->rows; y++)
                for (int x = (y + colour) & 1; x < d->cols; x += 2)
                    go_sweep_pixel(d, x, y, norm4, cost, phase, order[k]);
        }
    } else {
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < d->rows; y++)
            for (int x = (y + colour) & 1; x < d->cols; x += 2)
                go_sweep_pixel(d, x, y, norm4, cost, phase, stages);
    }
    return 0;
}

/* one colour kernel restricted to the rows [y0, y1): reads the whole frame's state, writes only
 * those rows.  A pixel's update depends only on the state BEFORE the launch (its own and the other
 * colour's), so running this on the state a device had before a launch and comparing the band with
 * the device's state after it is an exact check of that launch at any frame size. */
int gipuma_oracle_sweep_band(const gipuma_hip_desc *d, float *norm4, float *cost, int iteration,
                             int colour, unsigned stages, int y0, int y1)
{
    int rc = go_check(d);
    if (rc) return rc;
    if (y0 < 0 || y1 > d->rows || y0 > y1) return GIPUMA_HIP_ERR_ARG;
    const uint32_t phase = go_phase(iteration, colour);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; y++)
        for (int x = (y + colour) & 1; x < d->cols; x += 2)
            go_sweep_pixel(d, x, y, norm4, cost, phase, stages);
    return 0;
}

/* gipuma_compute_disp, gipuma.cu:1080-1103 */
int gipuma_oracle_finalize(const gipuma_hip_desc *d, float *norm4, const float *cost)
{
    int rc = go_check(d);
    if (rc) return rc;
    const gipuma_hip_camera *cam = &d->cameras[0];
#pragma omp parallel for
    for (int y = 0; y < d->rows; y++)
        for (int x = 0; x < d->cols; x++) {
            const size_t center = (size_t)y * (size_t)d->cols + (size_t)x;
            float pl[4], out[4];
            memcpy(pl, norm4 + 4 * center, sizeof pl);
            go_matvec(cam->R_orig_inv, pl, out);
            if (cost[center] != GO_MAXCOST)
                out[3] = go_depth_from_plane(cam, pl, x, y);
            else
                out[3] = 0;
            memcpy(norm4 + 4 * center, out, sizeof out);
        }
    return 0;
}

int gipuma_oracle_eval_cost(const gipuma_hip_desc *d, const float *planes, float *cost_out)
{
    int rc = go_check(d);
    if (rc) return rc;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < d->rows; y++)
        for (int x = 0; x < d->cols; x++) {
            const size_t center = (size_t)y * (size_t)d->cols + (size_t)x;
            cost_out[center] = go_multiview_cost(d, x, y, planes + 4 * center);
        }
    return 0;
}

/* gipuma<T>(), gipuma.cu:1906-1944 */
int gipuma_oracle_run(const gipuma_hip_desc *d, float *norm4, float *cost, int unfused)
{
    int rc = gipuma_oracle_init_planes(d, norm4, cost);
    if (rc) return rc;
    for (int it = 0; it < d->params.iterations; it++) {
        gipuma_oracle_sweep(d, norm4, cost, it, GIPUMA_BLACK, GIPUMA_STAGE_ALL, unfused);
        gipuma_oracle_sweep(d, norm4, cost, it, GIPUMA_RED, GIPUMA_STAGE_ALL, unfused);
    }
    return gipuma_oracle_finalize(d, norm4, cost);
}

static double go_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int gipuma_oracle_time(const gipuma_hip_desc *d, int n_iter_timed, double *sec_init,
                       double *sec_sweeps)
{
    int rc = go_check(d);
    if (rc) return rc;
    const size_t np = (size_t)d->rows * (size_t)d->cols;
    float *norm4 = (float *)malloc(np * 4 * sizeof(float));
    float *cost = (float *)malloc(np * sizeof(float));
    if (!norm4 || !cost) {
        free(norm4);
        free(cost);
        return GIPUMA_HIP_ERR_ARG;
    }
    double t0 = go_now();
    gipuma_oracle_init_planes(d, norm4, cost);
    double t1 = go_now();
    for (int it = 0; it < n_iter_timed; it++) {
        gipuma_oracle_sweep(d, norm4, cost, it, GIPUMA_BLACK, GIPUMA_STAGE_ALL, 0);
        gipuma_oracle_sweep(d, norm4, cost, it, GIPUMA_RED, GIPUMA_STAGE_ALL, 0);
    }
    double t2 = go_now();
    if (sec_init) *sec_init = t1 - t0;
    if (sec_sweeps) *sec_sweeps = t2 - t1;
    free(norm4);
    free(cost);
    return 0;
}

/* cpu_baseline leg of bench.py on a BOUNDED sample: init for rows [y0-5, y1+5), then one
 * iteration (black + red, all stages) over rows [y0, y1) only.  Work per pixel is the same for
 * every row and every iteration, so full-frame time = rows/(y1-y0) *
 * (sec_init_band + iterations * sec_iter_band). */
int gipuma_oracle_time_band(const gipuma_hip_desc *d, int y0, int y1, double *sec_init_band,
                            double *sec_iter_band)
{
    int rc = go_check(d);
    if (rc) return rc;
    if (y0 < 0 || y1 > d->rows || y0 >= y1) return GIPUMA_HIP_ERR_ARG;
    const size_t np = (size_t)d->rows * (size_t)d->cols;
    float *norm4 = (float *)calloc(np * 4, sizeof(float));
    float *cost = (float *)calloc(np, sizeof(float));
    if (!norm4 || !cost) {
        free(norm4);
        free(cost);
        return GIPUMA_HIP_ERR_ARG;
    }
    const int ya = y0 - 5 < 0 ? 0 : y0 - 5, yb = y1 + 5 > d->rows ? d->rows : y1 + 5;
    /* halo rows (untimed) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = ya; y < yb; y++)
        if (y < y0 || y >= y1)
            for (int x = 0; x < d->cols; x++) go_init_pixel(d, x, y, norm4, cost);
    double t0 = go_now();
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < d->cols; x++) go_init_pixel(d, x, y, norm4, cost);
    double t1 = go_now();
    for (int colour = 0; colour < 2; colour++) {
        const uint32_t phase = go_phase(0, colour);
#pragma omp parallel for schedule(dynamic, 1)
        for (int y = y0; y < y1; y++)
            for (int x = (y + colour) & 1; x < d->cols; x += 2)
                go_sweep_pixel(d, x, y, norm4, cost, phase, GIPUMA_STAGE_ALL);
    }
    double t2 = go_now();
    if (sec_init_band) *sec_init_band = t1 - t0;
    if (sec_iter_band) *sec_iter_band = t2 - t1;
    free(norm4);
    free(cost);
    return 0;
}

int gipuma_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the OpenMP loops from now on (the test harness passes the cores the cgroup grants) */
void gipuma_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
