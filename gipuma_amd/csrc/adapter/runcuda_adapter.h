/*
 * runcuda_adapter.h -- the reference's device entry point on top of the C-ABI.
 *
 *     int runcuda(GlobalState &gs);        reference gipuma.h:2, gipuma.cu:1962-1970
 *
 * Written as a template over the GlobalState type so the same unpacking code serves the
 * reference's real class (runcuda.cpp, compiled against /root/reference/globalstate.h) and any
 * mirror of it.  It reads exactly the fields the reference's kernels read:
 *   gs.params->{box_hsize, box_vsize, iterations, alpha, tau_color, tau_gradient, gamma,
 *               min_disparity, max_disparity, n_best, cost_comb, good_factor, color_processing}
 *   gs.cameras->{cols, rows, f, viewSelectionSubset[], viewSelectionSubsetNumber,
 *                cameras[i].{K, K_inv, R, t4, M_inv, P_col34, C4, R_orig_inv, fx, fy, f, alpha,
 *                            baseline, depthMin, depthMax}}
 *   gs.imgs[i]   (handles made by cudaCreateTextureObject, resolved to linear device buffers)
 *   gs.lines->{norm4, c}  in/out, host-visible on return (main.cpp:976-985)
 * and copies the <= 33 used cameras into one POD block once, instead of letting kernels chase
 * the ~3600 managed allocations of CameraParameters_cu (camera.h:45-51).
 */
#ifndef GIPUMA_RUNCUDA_ADAPTER_H
#define GIPUMA_RUNCUDA_ADAPTER_H

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../../include/gipuma_hip.h"

namespace gipuma_amd {

template <class GS>
int runcuda_impl(GS &gs, unsigned seed = 1)
{
    const auto &ap = *gs.params;
    auto &cp = *gs.cameras;
    const int channels = ap.color_processing ? 4 : 1; /* T = float4 : float, gipuma.cu:1965-1968 */
    const int n_sel = cp.viewSelectionSubsetNumber;
    int n_images = 1;
    for (int i = 0; i < n_sel; i++)
        if (cp.viewSelectionSubset[i] + 1 > n_images) n_images = cp.viewSelectionSubset[i] + 1;

    std::vector<gipuma_hip_camera> cams(n_images);
    std::vector<const float *> imgs(n_images, nullptr);
    int pitch = cp.cols * channels;
    for (int i = 0; i < n_images; i++) {
        const auto &c = cp.cameras[i];
        gipuma_hip_camera &o = cams[i];
        for (int k = 0; k < 9; k++) {
            o.K[k] = c.K[k];
            o.K_inv[k] = c.K_inv[k];
            o.R[k] = c.R[k];
            o.M_inv[k] = c.M_inv[k];
            o.R_orig_inv[k] = c.R_orig_inv[k];
        }
        o.t[0] = c.t4.x; o.t[1] = c.t4.y; o.t[2] = c.t4.z;
        o.P_col34[0] = c.P_col34.x; o.P_col34[1] = c.P_col34.y; o.P_col34[2] = c.P_col34.z;
        o.C[0] = c.C4.x; o.C[1] = c.C4.y; o.C[2] = c.C4.z;
        o.fx = c.fx; o.fy = c.fy; o.f = c.f; o.alpha = c.alpha; o.baseline = c.baseline;
        o.depth_min = c.depthMin; o.depth_max = c.depthMax;
        const cudaArray *a = gipuma_compat_texture(gs.imgs[i]);
        if (a) {
            imgs[i] = (const float *)a->data;
            pitch = (int)(a->pitch_bytes / sizeof(float));
        }
    }
    cams[0].f = cp.f; /* CameraParameters_cu::f is what getRndDispAndUnitVector_cu reads, gipuma.cu:904 */

    gipuma_hip_desc d{};
    d.abi_version = GIPUMA_HIP_ABI_VERSION;
    d.rows = cp.rows;
    d.cols = cp.cols;
    d.channels = channels;
    d.pitch = pitch;
    d.n_images = n_images;
    d.images = imgs.data();
    d.cameras = cams.data();
    d.n_selected = n_sel;
    d.selected = cp.viewSelectionSubset;
    d.params.box_hsize = ap.box_hsize;
    d.params.box_vsize = ap.box_vsize;
    d.params.iterations = ap.iterations;
    d.params.n_best = ap.n_best;
    d.params.cost_comb = ap.cost_comb;
    d.params.alpha = ap.alpha;
    d.params.tau_color = ap.tau_color;
    d.params.tau_gradient = ap.tau_gradient;
    d.params.gamma = ap.gamma;
    d.params.min_disparity = ap.min_disparity;
    d.params.max_disparity = ap.max_disparity;
    d.params.good_factor = ap.good_factor;
    d.seed = seed;
    int dev = 0;
    (void)hipGetDevice(&dev);
    d.device_id = dev;
    d.flags = GIPUMA_HIP_FLAG_IMAGES_ON_DEVICE;
    /* GlobalState has no field for it and main.cpp is to stay unchanged: the tolerance-judged mode (include/gipuma_hip.h,
     * the counterpart of the reference's --use_fast_math build, CMakeLists.txt:23) is chosen like the seed, by the environment */
    if (const char *fm = getenv("GIPUMA_FAST"))
        if (atoi(fm) != 0) d.flags |= GIPUMA_HIP_FLAG_FAST;
    /* ... and the reference-order validation mode (bit-identical to the reference's own arithmetic under fp32 filter weights) */
    if (const char *lm = getenv("GIPUMA_LITERAL"))
        if (atoi(lm) != 0) d.flags |= GIPUMA_HIP_FLAG_LITERAL;

    /* the lines the reference prints (gipuma.cu:1899-1912, 1952); scripts grep them */
    printf("Blocksize is %dx%d\n", ap.box_hsize, ap.box_vsize);
    printf("Number of iterations is %d\n", ap.iterations);
    gipuma_hip_timing t{};
    const int rc = gipuma_hip_run(&d, (float *)gs.lines->norm4, gs.lines->c, &t);
    if (rc != 0) { /* checkCudaErrors semantics, helper_cuda.h:890-905 */
        fprintf(stderr, "gipuma_hip: %s\n", gipuma_hip_last_error());
        exit(EXIT_FAILURE);
    }
    printf("\t\tTotal time needed for computation: %f seconds\n", (t.ms_sweeps + t.ms_finalize) / 1000.f);
    return 0;
}

}  // namespace gipuma_amd
#endif
