/*
 * gipuma_hip.h -- C-ABI of the MI355X-native PatchMatch multi-view stereo hot path.
 *
 * This is the drop-in boundary for the one entry point of the reference's device path,
 *
 *     int runcuda(GlobalState &gs);            (reference gipuma.h:2, gipuma.cu:1962-1970)
 *
 * restated as a plain-C interface: POD structs, raw pointers and sizes, no C++/torch/OpenCV
 * types.  Every field below names the reference field it carries (file:line).  The adapter that
 * keeps the reference's C++ signature on top of this ABI lives in
 * gipuma_amd/csrc/adapter/ (see INTEGRATION.md).
 *
 * Conventions
 *   - 3x3 matrices are row-major float[9] (the reference stores them row-major in 16-float
 *     buffers, config.h:150-241 / cameraGeometryUtils.h:160-167).
 *   - images are row-major float32, one value per pixel (gray, channels == 1) holding 0..255
 *     (main.cpp:941), `pitch` elements per row; image 0 is the reference view (config.h:21).
 *   - state planes: norm4[y*cols+x] = (nx, ny, nz, d) with n.X + d = 0 in reference-camera
 *     coordinates (linestate.h:10); cost[y*cols+x] is the aggregated multi-view cost
 *     (linestate.h:11).  After gipuma_hip_finalize / gipuma_hip_run, norm4 holds
 *     (n_world.xyz, depth) exactly like gipuma_compute_disp leaves it (gipuma.cu:1080-1103).
 *   - all entry points return 0 on success or a negative gipuma_hip_status; they never exit
 *     the process (the reference's checkCudaErrors does, helper_cuda.h:890-905).
 */
#ifndef GIPUMA_HIP_H
#define GIPUMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIPUMA_HIP_ABI_VERSION 1
#define GIPUMA_HIP_MAX_VIEWS 32 /* float costVector[32], gipuma.cu:736 */
#define GIPUMA_HIP_MAXCOST 1000.0f /* config.h:22 */

typedef enum {
    GIPUMA_HIP_OK = 0,
    GIPUMA_HIP_ERR_ARG = -1,      /* malformed descriptor */
    GIPUMA_HIP_ERR_DEVICE = -2,   /* HIP runtime error (text in gipuma_hip_last_error) */
    GIPUMA_HIP_ERR_NO_DEVICE = -3,/* no gfx950 device / extension not usable */
    GIPUMA_HIP_ERR_UNSUPPORTED = -4
} gipuma_hip_status;

/* cost combination, algorithmparameters.h:17 */
enum { GIPUMA_COMB_ALL = 0, GIPUMA_COMB_BEST_N = 1, GIPUMA_COMB_ANGLE = 2, GIPUMA_COMB_GOOD = 3 };

/* stages of one checkerboard half-sweep (gipuma.cu:1915-1935) */
enum {
    GIPUMA_STAGE_CLOSE = 1,  /* gipuma_*_spatialPropClose_cu, gipuma.cu:1471-1588 */
    GIPUMA_STAGE_FAR = 2,    /* gipuma_*_spatialPropFar_cu,   gipuma.cu:1353-1468 */
    GIPUMA_STAGE_REFINE = 4, /* gipuma_*_planeRefine_cu,      gipuma.cu:1590-1711 */
    GIPUMA_STAGE_ALL = 7
};
enum { GIPUMA_BLACK = 0 /* (x+y) even, gipuma.cu:1730-1734 */, GIPUMA_RED = 1 };

/* One calibrated view.  Mirrors Camera_cu (camera.h:7-62) after getCameraParameters()
 * (cameraGeometryUtils.h:311-345) has re-expressed every pose relative to the reference
 * camera, so that for view 0:  R = I, t = 0. */
typedef struct gipuma_hip_camera {
    float K[9];          /* Camera_cu::K      (per-view intrinsics, cameraGeometryUtils.h:311) */
    float K_inv[9];      /* Camera_cu::K_inv  (used for the reference view only, gipuma.cu:605) */
    float R[9];          /* Camera_cu::R      (relative rotation) */
    float t[3];          /* Camera_cu::t4 */
    float M_inv[9];      /* Camera_cu::M_inv  = inverse of P'[:, :3]; reference view only */
    float P_col34[3];    /* Camera_cu::P_col34 = P'[:, 3];           reference view only */
    float C[3];          /* Camera_cu::C4 camera centre;             reference view only */
    float R_orig_inv[9]; /* Camera_cu::R_orig_inv, world<-camera;    reference view only (gipuma.cu:1095) */
    float fx, fy;        /* Camera_cu::fx, fy  (of K_0, cameraGeometryUtils.h:314-323) */
    float f;             /* CameraParameters_cu::f / Camera_cu::f */
    float alpha;         /* Camera_cu::alpha = fx / fy */
    float baseline;      /* Camera_cu::baseline (constant 0.54, cameraGeometryUtils.h:305) */
    float depth_min;     /* Camera_cu::depthMin (main.cpp:902) */
    float depth_max;     /* Camera_cu::depthMax (main.cpp:903) */
} gipuma_hip_camera;

/* The subset of AlgorithmParameters (algorithmparameters.h:52-84) the device path reads. */
typedef struct gipuma_hip_params {
    int32_t box_hsize;     /* odd, --blocksize= */
    int32_t box_vsize;
    int32_t iterations;
    int32_t n_best;
    int32_t cost_comb;     /* GIPUMA_COMB_* */
    float alpha;           /* cost_alpha */
    float tau_color;
    float tau_gradient;
    float gamma;
    float min_disparity;   /* = f*baseline/depth_max (main.cpp:905) */
    float max_disparity;   /* = f*baseline/depth_min (main.cpp:906) */
    float good_factor;
} gipuma_hip_params;

/* images[] are device pointers (already resident).  The planes must be COMPLETE when gipuma_hip_create / gipuma_hip_run is
 * called: the library reads them on its own stream (desc.stream, or a non-blocking one it creates), which does not wait for
 * the stream that wrote them -- synchronise that stream (or pass it as desc.stream) first.  A plane read half-written fails
 * the 8-bit test and sends the session down the float kernels: same results once the plane is complete, far slower. */
#define GIPUMA_HIP_FLAG_IMAGES_ON_DEVICE 1u
#define GIPUMA_HIP_FLAG_UNFUSED 2u          /* run close/far/refine as 3 launches like the reference */
/* With IMAGES_ON_DEVICE: the library may keep what it derives from an image plane (the 8-bit check and the
 * window-packed copy the kernels sample) in a process-wide cache keyed by the plane's device address and
 * geometry, and reuse it in later sessions -- a scan's images serve as source view of many reference
 * views (the reference re-uploads every image for every view, main.cpp:960-968).  The caller promises not
 * to change or free those planes before gipuma_hip_cache_clear(). */
#define GIPUMA_HIP_FLAG_CACHE_IMAGES 4u
/* Mode flag (SURVEY.md 8b "mode flags (bit-exact/fast)"): without it every result is bit-identical to the CPU restatement of
 * the stated numerical model (since round 6 correctly rounded x/z, y/z and unfused multiply-adds like the
 * reference's source; the model's bilinear taps).  With it the session runs the TOLERANCE-JUDGED flavour of the same kernels
 * -- the operation-order freedoms the reference takes by being built with --use_fast_math (CMakeLists.txt:23) and nvcc's
 * contraction: the numerical model of rounds 1-5 (x * (1/z) for x / z, fused multiply-adds in the sample loop) with the
 * hardware reciprocal without its correcting step, no proof that the window's denominators are in the exact reciprocal's
 * range, and the nine divisions by the plane offset in getHomography_cu (gipuma.cu:339-356) as one reciprocal and a Markstein
 * step each.  (Measured and rejected, compiled only by A/B builds: a host-folded homography, tree sums -- pm_core.h.)  Same
 * algorithm, schedule and random numbers; results are judged by the fraction of pixels inside 1e-4 relative depth / 1e-3
 * normal of the default mode's and of the reference's (tests/test_fast_mode.py, tests/test_headline_parity.py,
 * DESIGN.md 3a), not bit for bit. */
#define GIPUMA_HIP_FLAG_FAST 8u
/* Mode flag: the REFERENCE-ORDER flavour.  The per-sample arithmetic of the patch cost in the literal operation order of the
 * reference's source -- one bilinear fetch per tap at the coordinates gipuma.cu:251-253 writes, each with its own fraction;
 * correctly rounded x/z and y/z (config.h:44-47); unfused multiply-adds (config.h:150-162, gipuma.cu:272-274, 672).  Results
 * equal the reference's OWN device code (compiled for the CPU with fp32 texture-filter weights, oracle/_ref;
 * tests/golden/ref_*.npz) in every bit of every plane and cost -- up to BASELINE's headline frame (tests/test_headline_parity.py).
 * Since round 6 it runs through the same kernels and schedule as the default mode (packed 8-bit windows, push / column-per-lane
 * / plane-keyed propagation, bounded refinement): about 1.3x the default mode's time.  The default mode differs from it in the
 * taps only (one window, the centre tap's fractions, differences taken on the texels).  Gray and colour (T = float4: the
 * reference's float4 operators and l1_norm, vector_operations.h, gipuma.cu:174-179); excludes GIPUMA_HIP_FLAG_FAST. */
#define GIPUMA_HIP_FLAG_LITERAL 16u

/* Everything runcuda() reads out of GlobalState (globalstate.h:24-45). */
typedef struct gipuma_hip_desc {
    uint32_t abi_version;            /* GIPUMA_HIP_ABI_VERSION */
    int32_t rows, cols;              /* CameraParameters_cu::rows, cols */
    int32_t channels;                /* 1 = gray (T=float, one float per pixel); 4 = colour (T=float4: B, G, R, unused) */
    int32_t pitch;                   /* elements per image row (>= cols*channels) */
    int32_t n_images;                /* reference + source views handed over (<= 512, config.h:2) */
    const float *const *images;      /* GlobalState::imgs[] as linear buffers (no texture HW on gfx950) */
    const gipuma_hip_camera *cameras;/* n_images entries, CameraParameters_cu::cameras[] */
    int32_t n_selected;              /* CameraParameters_cu::viewSelectionSubsetNumber (<= 32) */
    const int32_t *selected;         /* CameraParameters_cu::viewSelectionSubset[], indices into images[] */
    gipuma_hip_params params;        /* GlobalState::params */
    uint32_t seed;                   /* solver seed (extension: the reference seeds from clock64(), gipuma.cu:1019) */
    int32_t device_id;               /* HIP device ordinal */
    void *stream;                    /* hipStream_t to launch on, NULL = the library's own stream */
    uint32_t flags;                  /* GIPUMA_HIP_FLAG_* */
} gipuma_hip_desc;

/* Device-side timings of the last gipuma_hip_run (hipEvent pairs on the launch stream). */
typedef struct gipuma_hip_timing {
    float ms_init;     /* gipuma_init_cu2 */
    float ms_sweeps;   /* all red/black launches (the reference's own timed region minus finalize) */
    float ms_finalize; /* gipuma_compute_disp */
    float ms_total;    /* init + sweeps + finalize */
    int32_t n_sweep_launches;
    float ms_sweep_avg; /* ms_sweeps / n_sweep_launches: the dominant kernel's mean launch time */
} gipuma_hip_timing;

typedef struct gipuma_hip_session gipuma_hip_session;

/* ---- library ---- */
int gipuma_hip_version(void);                 /* GIPUMA_HIP_ABI_VERSION of the built library */
const char *gipuma_hip_last_error(void);      /* thread-local text of the last failure */
int gipuma_hip_device_count(void);            /* usable HIP devices (0 if none) */
/* frees everything kept for GIPUMA_HIP_FLAG_CACHE_IMAGES.  Sessions that use a cached packed image hold a use
 * count on it: while one of them is alive the call frees nothing and returns GIPUMA_HIP_ERR_ARG. */
int gipuma_hip_cache_clear(void);

/* Device self-test of an arithmetic shortcut the kernels rely on: v_rcp_f32 + one Newton step must
 * equal the IEEE-correct 1.0f/z bit for bit for EVERY float with biased exponent 1..252.  Runs the
 * exhaustive comparison (2^32 inputs, ~1 s) on `device_id` and returns the number of mismatches in
 * that range through *mismatches (0 expected); the gpu tests call it. */
int gipuma_hip_selftest_reciprocal(int device_id, unsigned long long *mismatches);
/* The default and the reference-order flavour form x / z, y / z of the warped point (vecdiv4, /root/reference/config.h:44-47;
 * getCorrespondingPoint_cu, gipuma.cu:207-217) as  r = RN(1/z), q = RN(x r), q' = RN(q + RN(x - q z) r)  where the window's
 * operands are provably in range -- the correctly rounded IEEE quotient for every pair of fp32 significands.  This runs the
 * proof by exhaustion for the denominators with significand bits z_first .. z_first + z_count - 1 (of 2^23) against all 2^23
 * numerators and returns the number of pairs whose result differs from the IEEE division (0 expected; the whole range takes
 * about a minute on an MI355X: profiles/r06_selftest_quotient.txt; the gpu tests run a slice). */
int gipuma_hip_selftest_quotient(int device_id, unsigned z_first, unsigned z_count, unsigned long long *mismatches);

/* ---- session: the pieces of gipuma<T>() (gipuma.cu:1825-1960), one call per launch ---- */
/* validates the descriptor, uploads/binds images and cameras, allocates norm4/cost in HBM
 * (replaces gs.lines->resize, linestate.h:16-24, and the setup half of gipuma<T>(), :1840-1861) */
int gipuma_hip_create(const gipuma_hip_desc *desc, gipuma_hip_session **out);
int gipuma_hip_destroy(gipuma_hip_session *s);
/* random plane per pixel + its cost: gipuma_init_cu2<float>, gipuma.cu:996-1051, launch :1906 */
int gipuma_hip_init_planes(gipuma_hip_session *s);
/* one colour of one iteration: the launches at gipuma.cu:1915-1923 (black) / :1927-1935 (red).
 * `stages` is a mask of GIPUMA_STAGE_*; the stages run in the reference order close, far, refine. */
int gipuma_hip_sweep(gipuma_hip_session *s, int iteration, int colour, unsigned stages);
/* plane -> (world normal, depth): gipuma_compute_disp, gipuma.cu:1080-1103, launch :1944 */
int gipuma_hip_finalize(gipuma_hip_session *s);
/* multi-view cost of a GIVEN plane field (pmCostMultiview_cu, gipuma.cu:720-806; what the
 * unused gipuma_initial_cost kernel, :1052-1079, computes).  planes: rows*cols*4 host floats,
 * cost_out: rows*cols host floats. Does not touch the session state. */
int gipuma_hip_eval_cost(gipuma_hip_session *s, const float *planes_host, float *cost_out_host);
/* host copies of the state planes (blocking) */
int gipuma_hip_get_state(gipuma_hip_session *s, float *norm4_host, float *cost_host);
int gipuma_hip_set_state(gipuma_hip_session *s, const float *norm4_host, const float *cost_host);
/* device pointers of the state planes (for callers that keep results in HBM).  A caller that
 * WRITES the planes through these pointers must afterwards call
 * gipuma_hip_set_state(s, NULL, NULL): like any set_state it tells the session that the stored
 * costs are no longer known to be the costs of the stored planes and that its record of which
 * planes changed in the last half-sweeps is void (the sweep kernels skip candidates equal to a
 * pixel's own plane, and neighbours that did not change since the pixel last met them, only
 * while both are known). */
int gipuma_hip_state_device_ptrs(gipuma_hip_session *s, float **norm4_dev, float **cost_dev);
/* init + iterations x (black, red) + finalize on the session, timed with HIP events.
 * Does not synchronise the host unless `timing` is non-NULL. */
int gipuma_hip_solve(gipuma_hip_session *s, gipuma_hip_timing *timing);
/* Device time of every half-sweep (one colour of one iteration: the launches at gipuma.cu:1915-1923 or
 * :1927-1935) of the last gipuma_hip_solve that was given a `timing`, in launch order: up to `capacity`
 * values to ms_half_sweep, their number (2 x iterations) to *n_half_sweeps.  *n_pushed = how many leading
 * half-sweeps read their propagation costs from pm::push_kernel launches that are timed with them
 * (DESIGN.md 5); every later half-sweep is exactly one fused sweep launch.  (The reference times the whole loop
 * with one cudaEvent pair, gipuma.cu:1908-1952.)  Pointers may be NULL. */
int gipuma_hip_launch_times(gipuma_hip_session *s, float *ms_half_sweep, int capacity, int *n_half_sweeps,
                            int *n_pushed);
/* Of the same solve: per half-sweep the device time of the pm::group_kernel launch that evaluated its propagation
 * costs (0 where the half-sweep had none: the pushed ones and every half-sweep of a problem the kernel does not
 * serve), so that the fused sweep launch's own time is ms_half_sweep[i] - ms_group[i].  These launches replace the
 * cost evaluations of gipuma.cu:1437-1462 / :1571-1582.  Pointers may be NULL. */
int gipuma_hip_group_times(gipuma_hip_session *s, float *ms_group, int capacity, int *n_half_sweeps);
/* Which kernels a full solve of this session launches per half-sweep h = 2 * iteration + colour (performance only; the
 * results do not depend on it):  info[0] = half-sweeps h < info[0] read propagation costs pushed by pm::push_kernel;
 * info[1] = from half-sweep info[1] on the costs come from the plane-keyed evaluation (pm_group.h), -1: never;
 * info[2] = 1: that evaluation is fused with the sweep (one pm::sweep_group_kernel launch per half-sweep), 0: a
 * pm::group_kernel launch in front of the sweep launch;  info[3] = half-sweeps h < info[3] run the column-per-lane
 * sweep kernel, 0: none. */
int gipuma_hip_schedule(gipuma_hip_session *s, int info[4]);

/* ---- one-shot: the whole of runcuda() ---- */
/* norm4_out: rows*cols*4 host floats, cost_out: rows*cols host floats (either may be NULL).
 * Results are host-visible on return like the reference's managed memory after
 * cudaDeviceSynchronize (gipuma.cu:1945, main.cpp:976-985). */
int gipuma_hip_run(const gipuma_hip_desc *desc, float *norm4_out, float *cost_out,
                   gipuma_hip_timing *timing);

#ifdef __cplusplus
}
#endif
#endif /* GIPUMA_HIP_H */
