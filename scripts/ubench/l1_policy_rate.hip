// Micro-benchmark: does a cache-policy modifier on global_load_dwordx4 (nt / sc0 / sc1) change the
// vector-L1 cost of divergent (one 128-byte line per lane) or coherent window loads on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o l1_policy_rate l1_policy_rate.hip && ./l1_policy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LOAD4(MOD)                                                                              \
    asm volatile("global_load_dwordx4 %0, %4, %8 " MOD "\n"                                      \
                 "global_load_dwordx4 %1, %5, %8 " MOD "\n"                                      \
                 "global_load_dwordx4 %2, %6, %8 " MOD "\n"                                      \
                 "global_load_dwordx4 %3, %7, %8 " MOD "\n"                                      \
                 "s_waitcnt vmcnt(0)"                                                            \
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                                        \
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base)                                 \
                 : "memory")

template <int POLICY>
__global__ void k(const char *base, int row_bytes, int mode, int iters, uint32_t *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 4, cc = lane & 15;
    uint32_t off = mode == 0 ? (uint32_t)(r * row_bytes + cc * 8) : (uint32_t)(lane * 128 + ((lane * 20) & 112));
    off += wave * 4 * row_bytes;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        for (int u = 0; u < 2; u++) {
            const uint32_t s = (uint32_t)((i & 7) * 8) + (uint32_t)(u * 8 * row_bytes);
            const uint32_t o0 = off + s, o1 = o0 + 2 * row_bytes, o2 = o0 + 4 * row_bytes, o3 = o0 + 6 * row_bytes;
            u32x4 a, b, c, d;
            if (POLICY == 0) LOAD4("");
            if (POLICY == 1) LOAD4("nt");
            if (POLICY == 2) LOAD4("sc0");
            if (POLICY == 3) LOAD4("sc1");
            if (POLICY == 4) LOAD4("sc0 sc1");
            if (POLICY == 5) LOAD4("sc0 sc1 nt");
            if (POLICY == 6) LOAD4("sc1 nt");
            if (POLICY == 7) LOAD4("sc0 nt");
            acc += a.x ^ b.y ^ c.z ^ d.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int POLICY>
void run(const char *name, const char *buf, uint32_t *out)
{
    for (int mode = 0; mode < 2; mode++) {
        const int iters = 2048, blocks = 256 * 4, threads = 256;
        const int row_bytes = 1608 * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<POLICY>, dim3(blocks), dim3(threads), 0, 0, buf, row_bytes, mode, 8, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<POLICY>, dim3(blocks), dim3(threads), 0, 0, buf, row_bytes, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * 8 * 16);
        printf("%-12s %-10s %.2f ns per wave-load per CU (= %.1f clk @2.4GHz)\n", name,
               mode ? "divergent" : "coherent", ns, ns * 2.4);
    }
}

int main()
{
    char *buf;
    uint32_t *out;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 1, 64 << 20);
    hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0>("(default)", buf, out);
    run<1>("nt", buf, out);
    run<2>("sc0", buf, out);
    run<3>("sc1", buf, out);
    run<4>("sc0 sc1", buf, out);
    run<5>("sc0 sc1 nt", buf, out);
    run<6>("sc1 nt", buf, out);
    run<7>("sc0 nt", buf, out);
    return 0;
}
