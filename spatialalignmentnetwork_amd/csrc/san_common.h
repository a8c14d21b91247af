// Shared helpers for the gfx950 kernels of libsan_hip.so (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/san_hip.h"

#define SAN_WAVE 64

void san_set_error(const char* fmt, ...);

#define SAN_CHECK_ARG(cond, msg)                       \
    do {                                               \
        if (!(cond)) {                                 \
            san_set_error("%s: %s", __func__, msg);    \
            return SAN_E_ARG;                          \
        }                                              \
    } while (0)

#define SAN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            san_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return (int)e__;                                                      \
        }                                                                         \
    } while (0)

static inline int san_cdiv(int a, int b) { return (a + b - 1) / b; }

// lazy normalisation applied by every consumer: lrelu(scale*x + shift, slope)
__device__ __forceinline__ float san_act(float x, float sc, float sh, float slope) {
    float v = fmaf(x, sc, sh);
    return v >= 0.f ? v : v * slope;
}

// 64-lane butterfly sum (all lanes end with the total)
__device__ __forceinline__ float san_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double san_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
