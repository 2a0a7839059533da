/* CUDA-on-CPU shim for oracle/_ref: every CUDA header name the reference includes maps to one file */
#include "../ref_cuda_on_cpu.h"
