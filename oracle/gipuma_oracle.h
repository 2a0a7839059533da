/*
 * gipuma_oracle.h -- CPU oracle for the PatchMatch hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Nothing under gipuma_amd/ includes, links or calls it; the product path fails loudly when
 * the HIP library is missing instead of falling back to this code.
 *
 * It takes the same POD descriptor as the C-ABI (include/gipuma_hip.h) so a test can hand one
 * object to both sides, but operates on caller-owned HOST state planes.
 *
 * Pin status: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be built here as a whole (CUDA + OpenCV).  The restatement is
 * pinned (a) against analytic known-answer tests derived from the cited formulas
 * (tests/test_oracle_kat.py) and (b) against the reference's OWN device functions compiled for
 * the CPU by oracle/Makefile into oracle/_ref/ (tests/test_oracle_vs_ref.py, fixtures in
 * tests/golden/).  See DESIGN.md "Oracle" for what (b) does and does not cover.
 */
#ifndef GIPUMA_ORACLE_H
#define GIPUMA_ORACLE_H

#include "../include/gipuma_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* whole path == runcuda(): init, iterations x (black, red), finalize.  norm4: rows*cols*4,
 * cost: rows*cols host floats.  `unfused` != 0 runs close / far / refine as three full-image
 * passes like the reference's six launches per iteration; 0 runs them pixel by pixel. Both
 * orders give identical bits (checked by tests). */
int gipuma_oracle_run(const gipuma_hip_desc *d, float *norm4, float *cost, int unfused);
int gipuma_oracle_init_planes(const gipuma_hip_desc *d, float *norm4, float *cost);
int gipuma_oracle_sweep(const gipuma_hip_desc *d, float *norm4, float *cost, int iteration,
                        int colour, unsigned stages, int unfused);
/* the same colour kernel on the rows [y0, y1) only (teacher-forced checks at full frame size) */
int gipuma_oracle_sweep_band(const gipuma_hip_desc *d, float *norm4, float *cost, int iteration,
                             int colour, unsigned stages, int y0, int y1);
int gipuma_oracle_finalize(const gipuma_hip_desc *d, float *norm4, const float *cost);
int gipuma_oracle_eval_cost(const gipuma_hip_desc *d, const float *planes, float *cost_out);
/* run only `n_iter_timed` iterations after init (cpu_baseline leg of bench.py); returns seconds
 * spent in init / sweeps via the two out-params */
int gipuma_oracle_time(const gipuma_hip_desc *d, int n_iter_timed, double *sec_init,
                       double *sec_sweeps);
int gipuma_oracle_time_band(const gipuma_hip_desc *d, int y0, int y1, double *sec_init_band,
                            double *sec_iter_band);
int gipuma_oracle_num_threads(void);
/* Flavours of the per-sample arithmetic (see gipuma_oracle.c), gray and colour: bit 0 = one bilinear fetch per tap at the
 * coordinates the source writes (else M1), bit 1 = IEEE x/z, y/z (else M2), bit 2 = unfused multiply-adds (else M3).
 * 6 (DEFAULT since round 6) = what the kernels' default mode computes (pm_sample.h, PM_MODEL 6); 7 = the operation order of
 * the reference's source (GIPUMA_HIP_FLAG_LITERAL); 0 = the model of rounds 1-5 (what GIPUMA_HIP_FLAG_FAST approximates).
 * Process-wide; tests restore the default with set_flavour(-1). */
void gipuma_oracle_set_flavour(int mask);
int gipuma_oracle_get_flavour(void);
int gipuma_oracle_default_flavour(void); /* 6: what the kernels' default mode computes; set_flavour(-1) returns to it */
void gipuma_oracle_set_threads(int n);
/* the sample loop eight window rows at a time (AVX2; the same operations per sample, bit-identical: tests compare) or scalar */
void gipuma_oracle_set_simd(int on);
int gipuma_oracle_get_simd(void);

/* ---- unit pieces, exported for the known-answer tests ---- */
float gipuma_oracle_exp(float x);
float gipuma_oracle_uniform(uint32_t seed, uint32_t phase, uint32_t x, uint32_t y, uint32_t draw);
void gipuma_oracle_homography(const gipuma_hip_camera *ref, const gipuma_hip_camera *to,
                              const float n[3], float dpl, float H[9]);
void gipuma_oracle_sample5(const float *img, int rows, int cols, int pitch, float x, float y,
                           float out[5]);
/* the three quantities the cost reads from the five bilinear taps, in the model's operation order (M1): centre value,
 * I(x+1,y) - I(x-1,y), I(x,y+1) - I(x,y-1) */
void gipuma_oracle_taps3(const float *img, int rows, int cols, int pitch, float x, float y, float out[3]);
float gipuma_oracle_aggregate(const float *view_costs, int n, int cost_comb, int n_best,
                              float good_factor);
float gipuma_oracle_view_cost(const gipuma_hip_desc *d, int view, int x, int y,
                              const float plane[4]);
float gipuma_oracle_multiview_cost(const gipuma_hip_desc *d, int x, int y, const float plane[4]);
/* the push formulation of gipuma_amd/csrc/pm_push.h restated on the CPU: the costs of producer (nx, ny)'s
 * plane at its eight consumers from ONE evaluation of dis per view on the stencil their windows share;
 * out[c] == gipuma_oracle_multiview_cost(consumer c, plane) bit for bit (tests/test_oracle_kat.py) */
int gipuma_oracle_push_costs(const gipuma_hip_desc *d, int nx, int ny, const float plane[4], float out[8],
                             int valid[8]);
float gipuma_oracle_depth_from_plane(const gipuma_hip_camera *cam, const float plane[4], int x,
                                     int y);
float gipuma_oracle_plane_d(const gipuma_hip_camera *cam, const float n[3], int x, int y,
                            float depth);
void gipuma_oracle_view_vector(const gipuma_hip_camera *cam, int x, int y, float v[3]);
int gipuma_oracle_refine_schedule(float max_disparity, float *delta_z, float *delta_n, int cap);

#ifdef __cplusplus
}
#endif
#endif
