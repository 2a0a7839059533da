// Exhaustive check: for which inputs does a v_rcp_f32 + Newton sequence equal the IEEE-correct
// 1.0f/z (the compiler's div_scale/div_fmas/div_fixup expansion), bit for bit?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_exact rcp_exact.hip && ./rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float s1(float z)
{
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float s2(float z)
{
    const float r = s1(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// per biased exponent of z: mismatch counts of s1, s2 and of raw rcp
__global__ void sweep(unsigned long long *bad1, unsigned long long *bad2, unsigned long long *bad0)
{
    const uint32_t hi = blockIdx.x;  // 65536 blocks x 65536 values
    unsigned c0 = 0, c1 = 0, c2 = 0;
    uint32_t ex = 0;
    for (uint32_t lo = threadIdx.x; lo < 65536; lo += blockDim.x) {
        const uint32_t bits = (hi << 16) | lo;
        const float z = __uint_as_float(bits);
        ex = (bits >> 23) & 0xff;
        const float ref = 1.0f / z;
        const uint32_t rb = __float_as_uint(ref);
        const bool nan = ref != ref;
        c0 += !nan && __float_as_uint(__builtin_amdgcn_rcpf(z)) != rb;
        c1 += !nan && __float_as_uint(s1(z)) != rb;
        c2 += !nan && __float_as_uint(s2(z)) != rb;
    }
    if (c0) atomicAdd(&bad0[ex], (unsigned long long)c0);
    if (c1) atomicAdd(&bad1[ex], (unsigned long long)c1);
    if (c2) atomicAdd(&bad2[ex], (unsigned long long)c2);
}

int main()
{
    unsigned long long *d, h[768];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(sweep, dim3(65536), dim3(256), 0, 0, d, d + 256, d + 512);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    int lo1 = 999, hi1 = -1, lo2 = 999, hi2 = -1;
    for (int e = 0; e < 256; e++) {
        t1 += h[e]; t2 += h[256 + e]; t0 += h[512 + e];
    }
    printf("mismatches vs IEEE 1/z over all 2^32 inputs (NaN results excluded): rcp %llu, s1 %llu, s2 %llu\n", t0, t1, t2);
    printf("biased exponents with s1 mismatches:");
    for (int e = 0; e < 256; e++) if (h[e]) printf(" %d:%llu", e, h[e]);
    printf("\nbiased exponents with s2 mismatches:");
    for (int e = 0; e < 256; e++) if (h[256 + e]) printf(" %d:%llu", e, h[256 + e]);
    printf("\n");
    return 0;
}
