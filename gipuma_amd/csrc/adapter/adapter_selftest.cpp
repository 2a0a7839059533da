// adapter_selftest.cpp -- drives the reference-signature runcuda() the way the reference's
// main.cpp does, for the gpu-marked drop-in test (tests/test_adapter_gpu.py):
//   new GlobalState / AlgorithmParameters in managed memory   (main.cpp:829, 1211)
//   fill cameras, selected views, parameters                   (main.cpp:888-933)
//   upload float images as "textures": gray (addImageToTextureFloatGray, main.cpp:607-656) or,
//   with -color_processing, float4 BGRA texels (addImageToTextureFloatColor, main.cpp:560-605)
//   runcuda(*gs)                                               (main.cpp:973)
//   read gs->lines->norm4 / c back on the host                 (main.cpp:976-985)
// Compiled against the reference's headers; only the resulting .so travels.
#include "gipuma.h"

#include "../../../include/gipuma_hip.h"

extern "C" int gipuma_adapter_selftest(const gipuma_hip_desc *d, float *norm4_out, float *cost_out)
{
    if (!d || (d->flags & GIPUMA_HIP_FLAG_IMAGES_ON_DEVICE)) return -1;
    AlgorithmParameters *algParams = new AlgorithmParameters;
    GlobalState *gs = new GlobalState;
    AlgorithmParameters &p = *algParams;
    p.box_hsize = d->params.box_hsize; p.box_vsize = d->params.box_vsize;
    p.iterations = d->params.iterations; p.n_best = d->params.n_best; p.cost_comb = d->params.cost_comb;
    p.alpha = d->params.alpha; p.tau_color = d->params.tau_color; p.tau_gradient = d->params.tau_gradient;
    p.gamma = d->params.gamma; p.good_factor = d->params.good_factor;
    p.min_disparity = d->params.min_disparity; p.max_disparity = d->params.max_disparity;
    p.color_processing = d->channels == 4;  // runcuda() picks gipuma<float4>, gipuma.cu:1965-1968
    gs->params = algParams;
    CameraParameters_cu &cp = *gs->cameras;
    cp.cols = d->cols; cp.rows = d->rows; cp.f = d->cameras[0].f;
    gs->params->cols = d->cols; gs->params->rows = d->rows;
    cp.viewSelectionSubsetNumber = d->n_selected;
    for (int i = 0; i < d->n_selected; i++) cp.viewSelectionSubset[i] = d->selected[i];
    for (int i = 0; i < d->n_images; i++) {
        const gipuma_hip_camera &c = d->cameras[i];
        Camera_cu &cam = cp.cameras[i];
        for (int k = 0; k < 9; k++) {
            cam.K[k] = c.K[k]; cam.K_inv[k] = c.K_inv[k]; cam.R[k] = c.R[k];
            cam.M_inv[k] = c.M_inv[k]; cam.R_orig_inv[k] = c.R_orig_inv[k];
        }
        cam.t4 = make_float4(c.t[0], c.t[1], c.t[2], 0);
        cam.P_col34 = make_float4(c.P_col34[0], c.P_col34[1], c.P_col34[2], 0);
        cam.C4 = make_float4(c.C[0], c.C[1], c.C[2], 0);
        cam.fx = c.fx; cam.fy = c.fy; cam.f = c.f; cam.alpha = c.alpha; cam.baseline = c.baseline;
        cam.depthMin = c.depth_min; cam.depthMax = c.depth_max;
    }
    gs->lines->n = d->rows * d->cols;
    gs->lines->resize(d->rows * d->cols);
    gs->lines->s = d->cols;
    gs->lines->l = d->cols;
    // addImageToTextureFloatGray, main.cpp:607-656 / addImageToTextureFloatColor, main.cpp:560-605
    for (int i = 0; i < d->n_images; i++) {
        cudaChannelFormatDesc channelDesc = d->channels == 4
                                                ? cudaCreateChannelDesc<float4>()
                                                : cudaCreateChannelDesc(32, 0, 0, 0, cudaChannelFormatKindFloat);
        checkCudaErrors(cudaMallocArray(&gs->cuArray[i], &channelDesc, d->cols, d->rows));
        checkCudaErrors(cudaMemcpy2DToArray(gs->cuArray[i], 0, 0, d->images[i], (size_t)d->pitch * sizeof(float),
                                            d->cols * sizeof(float) * d->channels, d->rows, cudaMemcpyHostToDevice));
        struct cudaResourceDesc resDesc;
        memset(&resDesc, 0, sizeof(resDesc));
        resDesc.resType = cudaResourceTypeArray;
        resDesc.res.array.array = gs->cuArray[i];
        struct cudaTextureDesc texDesc;
        memset(&texDesc, 0, sizeof(texDesc));
        texDesc.addressMode[0] = cudaAddressModeWrap;
        texDesc.addressMode[1] = cudaAddressModeWrap;
        texDesc.filterMode = cudaFilterModeLinear;
        texDesc.readMode = cudaReadModeElementType;
        texDesc.normalizedCoords = 0;
        checkCudaErrors(cudaCreateTextureObject(&(gs->imgs[i]), &resDesc, &texDesc, NULL));
    }
    const int rc = runcuda(*gs);
    // main.cpp:976-985: the host reads the managed planes right after the call
    const size_t n = (size_t)d->rows * d->cols;
    for (size_t k = 0; k < n; k++) {
        const float4 v = gs->lines->norm4[k];
        norm4_out[4 * k + 0] = v.x; norm4_out[4 * k + 1] = v.y; norm4_out[4 * k + 2] = v.z; norm4_out[4 * k + 3] = v.w;
        cost_out[k] = gs->lines->c[k];
    }
    for (int i = 0; i < d->n_images; i++) {
        cudaDestroyTextureObject(gs->imgs[i]);
        cudaFreeArray(gs->cuArray[i]);
    }
    delete gs;
    delete algParams;
    return rc;
}
