// Micro-benchmark: cost in the vector L1 (TCP) of the window-load patterns of the PatchMatch
// loop on gfx950.  Every wave issues back-to-back loads whose addresses stay inside a small,
// cache-resident region; reported: clocks per wave-load per CU (4 SIMDs issue concurrently).
//   hipcc --offload-arch=gfx950 -O3 -o l1_window_rate l1_window_rate.hip && ./l1_window_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;

// mode: 0 = 16 B per lane, consecutive (16-byte aligned)       [ideal coalesced dwordx4]
//       1 = 16 B per lane, lane stride 8 B, 4 rows of 16 lanes [the checkerboard window pattern]
//       2 = like 1 but +4 B (only 4-byte aligned)
//       3 = 16 B per lane, every lane in its own 128 B line     [incoherent planes]
//       4 = like 1, lane stride 4 B (all pixels, not checkerboard)
//       5 = 16 B per lane, lane stride 16 B but 4 rows of 16    [aligned, 4 rows]
template <int W>  // bytes per lane: 16, 8 or 4
__global__ void k(const char *base, int row_bytes, int mode, int iters, uint32_t *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off;
    const int r = lane >> 4, c = lane & 15;
    switch (mode) {
    case 0: off = lane * 16; break;
    case 1: off = r * row_bytes + c * 8; break;
    case 2: off = r * row_bytes + c * 8 + 4; break;
    case 3: off = lane * 128 + ((lane * 20) & 124); break;
    case 4: off = r * row_bytes + c * 4; break;
    case 5: off = r * row_bytes + c * 16; break;
    case 6: off = lane * 128; break;        // own line, 16-byte aligned
    case 7: off = lane * 128 + 4; break;    // own line, straddles a 16-byte boundary
    case 8: off = lane * 128 + 56; break;   // own line, straddles the 64-byte boundary
    case 9: off = lane * 128 + 120; break;  // straddles two 128-byte lines
    case 10: off = lane * 64; break;        // own 64-byte half line, aligned
    case 11: off = lane * 64 + 4; break;
    case 12: off = lane * 32 + 4; break;    // 4 lanes per line, misaligned
    default: off = lane * 256 + 4; break;   // own line, every second line
    }
    off += wave * 4 * row_bytes;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t o = off + (uint32_t)(u * 2 * row_bytes) + (uint32_t)((i & 7) * 8);
            if (W == 16) {
                const u32x4_a4 v = *(const u32x4_a4 *)(base + o);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else if (W == 8) {
                const u32x2_a4 v = *(const u32x2_a4 *)(base + o);
                acc += v.x ^ v.y;
            } else {
                acc += *(const uint32_t *)(base + o);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int W>
void run(const char *name, const char *buf, uint32_t *out, int mode)
{
    const int iters = 2048, blocks = 256 * 4, threads = 256;  // 4 blocks (16 waves) per CU
    const int row_bytes = 1608 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(threads), 0, 0, buf, row_bytes, mode, 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(threads), 0, 0, buf, row_bytes, mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)iters * 8 * 16;  // wave-loads per CU (16 waves)
    const double ns = ms * 1e6 / loads_per_cu;
    printf("%-34s W=%2d  %.2f ns per wave-load per CU (= %.1f clk @2.4GHz)\n", name, W, ns, ns * 2.4);
}

int main()
{
    char *buf;
    uint32_t *out;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 1, 64 << 20);
    hipMalloc(&out, 256 * 4 * 256 * 4);
    run<16>("aligned consecutive", buf, out, 0);
    run<16>("window: stride 8 B, 4 rows", buf, out, 1);
    run<16>("window: stride 8 B + 4, 4 rows", buf, out, 2);
    run<16>("every lane own line", buf, out, 3);
    run<16>("window: stride 4 B, 4 rows", buf, out, 4);
    run<16>("aligned 16 B stride, 4 rows", buf, out, 5);
    run<16>("own line, aligned 16", buf, out, 6);
    run<16>("own line, +4 (straddle 16 B)", buf, out, 7);
    run<16>("own line, +56 (straddle 64 B)", buf, out, 8);
    run<16>("+120 (straddle two lines)", buf, out, 9);
    run<16>("own 64 B half line, aligned", buf, out, 10);
    run<16>("own 64 B half line, +4", buf, out, 11);
    run<16>("4 lanes per line, +4", buf, out, 12);
    run<16>("every second line, +4", buf, out, 13);
    run<4>("4 B: own line aligned", buf, out, 6);
    run<8>("8 B: own line +4", buf, out, 7);
    run<8>("8 B: stride 8 B, 4 rows", buf, out, 1);
    run<8>("8 B: every lane own line", buf, out, 3);
    run<4>("4 B: stride 8 B, 4 rows", buf, out, 1);
    run<4>("4 B: every lane own line", buf, out, 3);
    return 0;
}
