// Micro-benchmark: what does a DIVERGENT window load cost in the vector L1 of gfx950 when its lines are
// already resident (hits), compared with the same pattern missing to L2?  Two 16-byte loads per lane and
// step: the second one either into the line the first one touched (+8 B: a horizontally adjacent window)
// or into another line.
//   hipcc --offload-arch=gfx950 -O3 -o l1_hit_rate l1_hit_rate.hip && ./l1_hit_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;

// every lane walks its own sequence of lines: line index = (lane * 17 + step * stride_lines) % region_lines
// region_lines small -> the working set of the CU stays in L1; large -> every load misses to L2
// pair: 0 = one load per step; 1 = + a second load 8 B further (same line); 2 = + a second load in another line
__global__ void k(const char *base, int region_lines, int pair, int iters, uint32_t *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wave_base = (uint32_t)(blockIdx.x * 4 + wave) * (uint32_t)region_lines * 128u;
    uint32_t acc = 0;
    uint32_t line = (uint32_t)(lane * 17) % (uint32_t)region_lines;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t o = wave_base + line * 128u + 36u;
            const u32x4_a4 v = *(const u32x4_a4 *)(base + o);
            acc += v.x ^ v.y ^ v.z ^ v.w;
            if (pair == 1) {
                const u32x4_a4 w = *(const u32x4_a4 *)(base + o + 8);
                acc += w.x ^ w.y ^ w.z ^ w.w;
            } else if (pair == 2) {
                const uint32_t l2 = (line + 7u) % (uint32_t)region_lines;
                const u32x4_a4 w = *(const u32x4_a4 *)(base + wave_base + l2 * 128u + 36u);
                acc += w.x ^ w.y ^ w.z ^ w.w;
            }
            line += 67u;
            if (line >= (uint32_t)region_lines) line -= (uint32_t)region_lines;
            if (line >= (uint32_t)region_lines) line %= (uint32_t)region_lines;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

void run(const char *name, const char *buf, uint32_t *out, int region_lines, int pair, int waves_per_cu)
{
    const int iters = 1024, blocks = 256, threads = 64 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, region_lines, pair, 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, region_lines, pair, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_cu = (double)iters * 8 * waves_per_cu;  // wave-steps per CU
    const double ns = ms * 1e6 / steps_per_cu;
    printf("%-58s %2d waves/CU  %7.2f ns per wave-step per CU (= %6.1f clk @2.4GHz)\n", name, waves_per_cu, ns, ns * 2.4);
}

int main()
{
    char *buf;
    uint32_t *out;
    const size_t bytes = (size_t)1 << 30;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 1, bytes);
    hipMalloc(&out, 256 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        // 64 lines per wave = 8 KB: w waves keep 8w KB in the 32 KB L1
        run("divergent, L1 resident (64 lines per wave), 1 load", buf, out, 64, 0, w);
        run("divergent, L1 resident, + same-line load (+8 B)", buf, out, 64, 1, w);
        run("divergent, L1 resident, + other-line load", buf, out, 64, 2, w);
    }
    for (int w = 2; w <= 4; w *= 2) {
        // 8192 lines per wave = 1 MB: misses in L1, hits in L2
        run("divergent, L2 resident (8192 lines per wave), 1 load", buf, out, 8192, 0, w);
        run("divergent, L2 resident, + same-line load (+8 B)", buf, out, 8192, 1, w);
        run("divergent, L2 resident, + other-line load", buf, out, 8192, 2, w);
    }
    return 0;
}
