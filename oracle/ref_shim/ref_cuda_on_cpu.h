/*
 * ref_cuda_on_cpu.h -- just enough of the CUDA language and runtime, on the CPU, to compile the
 * DEVICE part of the reference's gipuma.cu (everything before its <<<...>>> launcher) with g++.
 * TEST INFRASTRUCTURE: used only by oracle/Makefile to build oracle/_ref/libgipuma_ref.so, the
 * reference's own arithmetic, against which the oracle restatement is pinned.
 *
 * What is the reference's and what is ours in that library:
 *   reference (compiled from /root/reference, untouched): every __device__ function and kernel
 *     body -- planes, homography, patch cost, view aggregation, propagation, refinement, final
 *     conversion -- and the GlobalState / Camera_cu / AlgorithmParameters structs.
 *   ours (this shim): the three things that are hardware or toolkit, not reference source:
 *     tex2D (texture unit), curand (toolkit RNG, and unseeded in the reference: SURVEY F2),
 *     expf/rsqrtf under --use_fast_math; plus the launch loop (ref_harness.cpp).
 */
#ifndef REF_CUDA_ON_CPU_H
#define REF_CUDA_ON_CPU_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <iostream>
#include <string>
#include <vector>

/* ---- language ---- */
#define __device__
#define __global__
#define __host__
/* a block's shared memory, its thread / block index and its barrier pass belong to the host thread that runs the
 * block: the harness runs the blocks of a launch under OpenMP (they are independent: a launch only writes pixels
 * of its own colour and reads the other's) */
#define __shared__ thread_local
#define __constant__
#define __managed__
#define __forceinline__ inline __attribute__((always_inline))
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(...)

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }

extern thread_local uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim; /* launch constants */
/* block-level barrier: the harness runs every block twice; in pass 0 each thread stops at the
 * barrier (after loading its slice of the shared tile), in pass 1 it runs through */
extern thread_local int ref_pass;
#define __syncthreads() do { if (ref_pass == 0) return; } while (0)

/* ---- math under --use_fast_math (numerical model M2) ---- */
float ref_model_expf(float x);
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __int2float_rn(int x) { return (float)x; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline long long clock64() { return 0; }
#define expf(x) ref_model_expf(x)

/* ---- runtime ---- */
typedef int cudaError_t;
#define cudaSuccess 0
typedef unsigned long long cudaTextureObject_t;
struct cudaArray;
#define checkCudaErrors(x) (x)
#define HELPER_CUDA_H /* keep the reference's vendored NVIDIA helper_cuda.h out */
template <class T> static inline cudaError_t cudaMallocManaged(T **p, size_t n) { *p = (T *)calloc(1, n); return 0; }
template <class T> static inline cudaError_t cudaMalloc(T **p, size_t n) { *p = (T *)calloc(1, n); return 0; }
static inline cudaError_t cudaFree(void *p) { free(p); return 0; }

/* ---- texture unit (numerical model M1) ---- */
struct RefTexture { const float *data; int cols, rows, pitch, channels; }; /* pitch in floats */
extern RefTexture ref_textures[];
extern int ref_tex_mode; /* 0: fp32 lerp weights; 1: weights rounded to 8 fractional bits like CUDA */
template <class T> T tex2D(cudaTextureObject_t tex, float x, float y);
template <> float tex2D<float>(cudaTextureObject_t tex, float x, float y);
template <> float4 tex2D<float4>(cudaTextureObject_t tex, float x, float y);

/* ---- cuRAND device API (numerical model M4) ---- */
struct curandState { unsigned x, y, n; unsigned pad[9]; }; /* 48 bytes like curandStateXORWOW */
extern unsigned ref_seed, ref_phase;
static inline void curand_init(long long seed, int sequence, int offset, curandState *s)
{
    (void)seed;
    s->y = (unsigned)sequence; /* the reference passes (clock64(), p.y, p.x), gipuma.cu:1019 */
    s->x = (unsigned)offset;
    s->n = 0;
}
float curand_uniform(curandState *s);

#endif
