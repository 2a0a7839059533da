// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the
// PatchMatch inner loop is made of, on gfx950.  One wave per SIMD and 4 waves per SIMD variants.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                         \
    __global__ void name(float *out, int iters, unsigned long long *clk)          \
    {                                                                              \
        float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f, e = 3.0f;   \
        float a2 = a + 1, b2 = b + 1, c2 = c + 1, d2 = d + 1;                     \
        unsigned u = threadIdx.x, v = 77u, w = 3u, u2 = u + 1;                    \
        /* shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around the loop */ \
        const unsigned long long s0 = __builtin_readcyclecounter(), r0 = wall_clock64(); \
        for (int i = 0; i < iters; i++) { REP16(body) }                           \
        const unsigned long long s1 = __builtin_readcyclecounter(), r1 = wall_clock64(); \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1 - s0; clk[1] = r1 - r0; } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + a2 + b2 + c2 + d2 + (float)(u + v + w + u2); \
    }

// 4 independent chains per body so latency does not bound
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(a2));)
KERNEL(k_fma_s, asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(2.5f), "v"(a2));)
KERNEL(k_fma_2, asm volatile("v_fma_f32 %0, %4, %4, %0\n v_fma_f32 %1, %4, %4, %1\n v_fma_f32 %2, %4, %4, %2\n v_fma_f32 %3, %4, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_add, asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b2));)
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1" : "+v"(*(double*)&a), "+v"(*(double*)&c) : "v"(*(double*)&a2), "v"(*(double*)&c2));)
KERNEL(k_pkadd, asm volatile("v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1\n v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1" : "+v"(*(double*)&a), "+v"(*(double*)&c) : "v"(*(double*)&a2));)
KERNEL(k_cvtub, asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));)
KERNEL(k_cvti, asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %4\n v_cvt_f32_i32 %2, %4\n v_cvt_f32_i32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));)
KERNEL(k_cvtf2i, asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %4\n v_cvt_i32_f32 %2, %4\n v_cvt_i32_f32 %3, %4" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(a));)
KERNEL(k_addu, asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %4, %0, %0\n v_mad_u32_u24 %1, %4, %1, %1\n v_mad_u32_u24 %2, %4, %2, %2\n v_mad_u32_u24 %3, %4, %3, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_sdwa, asm volatile("v_sub_u32_sdwa %0, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1\n v_sub_u32_sdwa %2, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2\n v_sub_u32_sdwa %3, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_min, asm volatile("v_min_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_max_f32 %2, %4, %2\n v_max_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_min64, asm volatile("v_min_f32_e64 %0, %4, %0\n v_min_f32_e64 %1, %4, %1\n v_max_f32_e64 %2, %4, %2\n v_max_f32_e64 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_add64, asm volatile("v_add_f32_e64 %0, %4, %0\n v_add_f32_e64 %1, %4, %1\n v_add_f32_e64 %2, %4, %2\n v_add_f32_e64 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_floor, asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %4, 16, 8\n v_bfe_u32 %2, %4, 8, 8\n v_bfe_u32 %3, %4, 16, 8" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(u));)
KERNEL(k_divfmas, asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(a2));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_fmamix, asm volatile("v_fma_f32 %0, %4, %5, %0\n v_add_u32 %2, %6, %2\n v_fma_f32 %1, %4, %5, %1\n v_add_u32 %3, %6, %3" : "+v"(a), "+v"(b), "+v"(u), "+v"(v) : "v"(e), "v"(a2), "v"(77));)
KERNEL(k_fmamixh, asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(u));)
KERNEL(k_fmamix3, asm volatile("v_fma_mix_f32 %0, %4, 1.0, -%5 op_sel:[1,0,0] op_sel_hi:[1,0,1]\n v_fma_mix_f32 %1, %4, 1.0, -%5 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n v_fma_mix_f32 %2, %4, 1.0, -%5 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n v_fma_mix_f32 %3, %4, 1.0, -%5 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(u), "v"(v));)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(a2));)
KERNEL(k_min3, asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(a2));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %4, %0, %5\n v_perm_b32 %1, %4, %1, %5\n v_perm_b32 %2, %4, %2, %5\n v_perm_b32 %3, %4, %3, %5" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(0x4b000000u), "s"(0x070c0c01u));)
KERNEL(k_andor, asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "s"(0xffu), "v"(0x4b000000u));)
KERNEL(k_and, asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "s"(0x3ffu));)
KERNEL(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_add_lshl_u32 %2, %2, %4, 2\n v_add_lshl_u32 %3, %3, %4, 2" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_cvtu32, asm volatile("v_cvt_u32_f32 %0, %4\n v_cvt_u32_f32 %1, %4\n v_cvt_u32_f32 %2, %4\n v_cvt_u32_f32 %3, %4" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(a));)
KERNEL(k_minu, asm volatile("v_min_u32 %0, %4, %0\n v_min_u32 %1, %4, %1\n v_min_u32 %2, %4, %2\n v_min_u32 %3, %4, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2) : "v"(77));)
KERNEL(k_subabs, asm volatile("v_sub_f32_e64 %0, |%0|, %4\n v_sub_f32_e64 %1, |%1|, %4\n v_sub_f32_e64 %2, |%2|, %4\n v_sub_f32_e64 %3, |%3|, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_fract, asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3" : "+v"(u), "+v"(v), "+v"(w), "+v"(u2));)
KERNEL(k_cvtf16, asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %4\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u));)

static unsigned long long *g_clk;  // device: [0] shader-clock ticks, [1] 100 MHz ticks of the timed launch

template <typename K>
void run(const char *name, K k, float *d, int waves_per_simd)
{
    const int iters = 4096, ops = iters * 16 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int block = 256 * waves_per_simd;  // 4 SIMDs x waves_per_simd waves per CU
    hipLaunchKernelGGL(k, dim3(256), dim3(block), 0, 0, d, 16, g_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(block), 0, 0, d, iters, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpy(clk, g_clk, sizeof clk, hipMemcpyDeviceToHost);
    // per SIMD: waves_per_simd waves each issuing `ops` instructions.  The shader clock the launch ran at is
    // measured by the kernel itself: s_memtime ticks per s_memrealtime tick (100 MHz)
    const double ns_per_inst = ms * 1e6 / ((double)ops * waves_per_simd);
    const double sclk_mhz = clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0;
    // clocks per instruction per SIMD from the event time and the clock the kernel itself saw.  (Wave 0's own elapsed
    // shader clocks per ITS instruction ride along: with several waves per SIMD the oldest wave keeps its
    // single-wave pace while the younger ones take what is left, so that figure is not the SIMD's issue rate.)
    const double clk_per_inst = ns_per_inst * sclk_mhz * 1e-3;
    const double clk_wave0 = (double)clk[0] / (double)ops;
    printf("%-12s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD = %.2f shader clocks at the sustained SCLK of %.0f MHz "
           "(wave 0 alone: %.2f clk per own instruction)\n", name, waves_per_simd, ns_per_inst, clk_per_inst, sclk_mhz, clk_wave0);
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipMalloc(&g_clk, 2 * sizeof(unsigned long long));
    printf("# encodings: fma VOP3; add/mul/min/max VOP2 (and _e64 = VOP3 forms of add, min); cvt_*/floor/fract/rcp/mov "
           "VOP1; med3/min3/bfe/perm/and_or/lshl_add/mad/div_fixup/fma_mix VOP3(P); sub_sdwa SDWA\n");
    for (int w : {1, 2, 4}) {
        run("fma", k_fma, d, w); run("fma_sgpr_src", k_fma_s, d, w); run("fma_2_vgprs", k_fma_2, d, w); run("add", k_add, d, w); run("add_e64", k_add64, d, w); run("mul", k_mul, d, w);
        run("pk_fma", k_pkfma, d, w); run("pk_add", k_pkadd, d, w);
        run("cvt_ubyte", k_cvtub, d, w); run("cvt_f32_i32", k_cvti, d, w); run("cvt_i32_f32", k_cvtf2i, d, w);
        run("add_u32", k_addu, d, w); run("mul_lo_u32", k_mullo, d, w); run("mad_u24", k_mad24, d, w);
        run("sub_sdwa", k_sdwa, d, w); run("min/max", k_min, d, w); run("min/max_e64", k_min64, d, w); run("rcp", k_rcp, d, w);
        run("floor", k_floor, d, w); run("bfe", k_bfe, d, w); run("div_fixup", k_divfmas, d, w);
        run("mov", k_mov, d, w); run("fma+addu", k_fmamix, d, w);
        run("fma_mix_h", k_fmamixh, d, w); run("fma_mix_sub", k_fmamix3, d, w); run("med3", k_med3, d, w);
        run("min3/max3", k_min3, d, w); run("perm", k_perm, d, w); run("and_or", k_andor, d, w); run("and", k_and, d, w);
        run("lshl_add", k_lshladd, d, w); run("cvt_u32_f32", k_cvtu32, d, w); run("min_u32", k_minu, d, w);
        run("sub_abs_e64", k_subabs, d, w); run("fract", k_fract, d, w); run("lshr/lshl", k_lshr, d, w);
        run("cvt_f32_f16", k_cvtf16, d, w);
    }
    return 0;
}
