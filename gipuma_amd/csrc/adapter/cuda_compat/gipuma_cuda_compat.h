/*
 * gipuma_cuda_compat.h -- the CUDA runtime names the reference's HOST code uses (main.cpp,
 * globalstate.h, camera.h, cameraparameters.h, linestate.h, managed.h), served by the HIP
 * runtime, so that those files compile unchanged with hipcc on an MI355X box.
 *
 * This is boundary glue for the reference's own sources, not part of the compute path: the
 * kernels in pm_device.h are written for gfx950 directly.  Two things differ from a plain
 * cuda->hip rename:
 *   * gfx950 has no image instructions, so "textures" are records of a linear
 *     device buffer: cudaMallocArray / cudaMemcpy2DToArray / cudaCreateTextureObject
 *     (main.cpp:607-656) allocate, fill and register such a buffer; runcuda() reads it back
 *     through gipuma_compat_texture().  Filtering/addressing flags are accepted and ignored --
 *     the path always samples with bilinear filtering and clamp-to-edge, which is what the
 *     reference's settings amount to.
 *   * checkCudaErrors keeps the reference's behaviour: print and exit(EXIT_FAILURE)
 *     (helper_cuda.h:890-905); the vendored NVIDIA helper_cuda.h itself is kept out.
 */
#ifndef GIPUMA_CUDA_COMPAT_H
#define GIPUMA_CUDA_COMPAT_H

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HELPER_CUDA_H /* skip /reference/helper_cuda.h (NVIDIA sample helpers) */

/* ---- errors ---- */
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
#define checkCudaErrors(val) gipuma_compat_check((val), #val, __FILE__, __LINE__)
static inline void gipuma_compat_check(hipError_t e, const char *what, const char *file, int line)
{
    if (e != hipSuccess) {
        fprintf(stderr, "CUDA error at %s:%d code=%d(%s) \"%s\" \n", file, line, (int)e, hipGetErrorString(e), what);
        exit(EXIT_FAILURE);
    }
}

/* ---- device management (main.cpp:658-692) ---- */
typedef hipDeviceProp_t cudaDeviceProp;
#define cudaGetDeviceCount hipGetDeviceCount
#define cudaGetDeviceProperties hipGetDeviceProperties
#define cudaSetDevice hipSetDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMemGetInfo hipMemGetInfo
#define cudaDeviceReset hipDeviceReset
enum cudaLimit { cudaLimitPrintfFifoSize = 1 };
static inline cudaError_t cudaDeviceSetLimit(cudaLimit, size_t) { return hipSuccess; }

/* ---- memory (managed.h, camera.h, linestate.h) ---- */
template <class T> static inline cudaError_t cudaMallocManaged(T **p, size_t n)
{
    return hipMallocManaged((void **)p, n, hipMemAttachGlobal);
}
template <class T> static inline cudaError_t cudaMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline cudaError_t cudaFree(void *p) { return hipFree(p); }
enum cudaMemcpyKind {
    cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2,
    cudaMemcpyDeviceToDevice = 3, cudaMemcpyDefault = 4
};

/* ---- cuRAND device state: only its size matters (globalstate.h:28; never initialised, F2) ---- */
struct curandState { unsigned int v[12]; };

/* ---- "textures": linear device buffers behind opaque handles ---- */
enum cudaChannelFormatKind { cudaChannelFormatKindSigned, cudaChannelFormatKindUnsigned, cudaChannelFormatKindFloat };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f)
{
    cudaChannelFormatDesc d = {x, y, z, w, f};
    return d;
}
template <class T> static inline cudaChannelFormatDesc cudaCreateChannelDesc();   /* main.cpp:567 */
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float>()
{
    return cudaCreateChannelDesc(32, 0, 0, 0, cudaChannelFormatKindFloat);
}
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float4>()
{
    return cudaCreateChannelDesc(32, 32, 32, 32, cudaChannelFormatKindFloat);
}
struct cudaArray {
    void *data;          /* device memory, row-major, pitch_bytes per row */
    size_t width, height, pitch_bytes;
    cudaChannelFormatDesc desc;
};
static inline cudaError_t cudaMallocArray(cudaArray **a, const cudaChannelFormatDesc *d, size_t w, size_t h)
{
    cudaArray *arr = new cudaArray;
    arr->desc = *d;
    arr->width = w;
    arr->height = h;
    arr->pitch_bytes = w * (size_t)((d->x + d->y + d->z + d->w) / 8);
    hipError_t e = hipMalloc(&arr->data, arr->pitch_bytes * h);
    if (e != hipSuccess) { delete arr; return e; }
    *a = arr;
    return hipSuccess;
}
static inline cudaError_t cudaFreeArray(cudaArray *a)
{
    if (!a) return hipSuccess;
    hipError_t e = hipFree(a->data);
    delete a;
    return e;
}
static inline cudaError_t cudaMemcpy2DToArray(cudaArray *dst, size_t wOffset, size_t hOffset, const void *src,
                                              size_t spitch, size_t width, size_t height, cudaMemcpyKind)
{
    char *d = (char *)dst->data + hOffset * dst->pitch_bytes + wOffset;
    return hipMemcpy2D(d, dst->pitch_bytes, src, spitch, width, height, hipMemcpyHostToDevice);
}
enum cudaResourceType { cudaResourceTypeArray = 0 };
struct cudaResourceDesc {
    cudaResourceType resType;
    struct { struct { cudaArray *array; } array; } res;
};
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1 };
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureReadMode { cudaReadModeElementType = 0 };
struct cudaTextureDesc {
    cudaTextureAddressMode addressMode[3];
    cudaTextureFilterMode filterMode;
    cudaTextureReadMode readMode;
    int sRGB;
    int normalizedCoords;
};
typedef unsigned long long cudaTextureObject_t;

/* one registry per process (function-local static in an inline function) */
inline std::vector<const cudaArray *> &gipuma_compat_registry()
{
    static std::vector<const cudaArray *> r(1, (const cudaArray *)nullptr); /* handle 0 = invalid */
    return r;
}
static inline cudaError_t cudaCreateTextureObject(cudaTextureObject_t *t, const cudaResourceDesc *res,
                                                  const cudaTextureDesc *, const void *)
{
    gipuma_compat_registry().push_back(res->res.array.array);
    *t = (cudaTextureObject_t)(gipuma_compat_registry().size() - 1);
    return hipSuccess;
}
static inline cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t)
{
    if (t < gipuma_compat_registry().size()) gipuma_compat_registry()[t] = nullptr;
    return hipSuccess;
}
static inline const cudaArray *gipuma_compat_texture(cudaTextureObject_t t)
{
    return t < gipuma_compat_registry().size() ? gipuma_compat_registry()[t] : nullptr;
}

#endif
