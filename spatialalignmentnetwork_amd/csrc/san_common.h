// Shared helpers for the gfx950 kernels of libsan_hip.so (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/san_hip.h"

#define SAN_WAVE 64

void san_set_error(const char* fmt, ...);

// 1x1 / transposed convolution as a one-stage GEMM (san_conv1x1.hip), launched by conv_bf16x3_run (san_conv_bf16.hip) for KS = 1
// layers whose weights are packed as two fp16 parts.  Internal to the library (not part of the ABI).
struct SanGemm1x1Args {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    const void* wp;            // packed image [chunk][block of 16 couts][part][64 lanes] x 16 B (ks = 1: one K-step per chunk)
    const float* bias;
    float* y;
    float* part;               // statistics records or null: `slots` per (sample, channel) plane (x 4 interleaved when shuffling)
    const uint32_t* amax;      // gradient input: its amax record (the input is scaled by a power of two), else null
    const float* w_tail;       // fp16-format weights: {S_w, 1 / S_w} behind the packed image (the accumulators get 1 / S_w), else null
    float in_slope;
    int x_ctot, x_coff, cin;
    int y_ctot, y_coff, cout;  // cout: channels of the GEMM (4 x the real channels when shuffling)
    int N, H, W;
    int chunks, nblkp;
    int shuffle;               // 1: ConvTranspose2d 2x2 s2 -- 4 virtual channels per real channel + pixel shuffle
    int bf1;                   // 1: bf16-format image, one part (plain bf16 operands, one product); 0: two fp16 parts, three products
    int slots;
    int ngrp, ptiles;          // (filled by the launcher)
};
bool san_gemm1x1_enabled();
int san_gemm1x1_f16_run(SanGemm1x1Args a, void* stream);

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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a launcher keeps one of these per kernel and asks
// before every launch (round 6, ADVICE r5: a process-wide `static bool` left the second GPU of one process unconfigured).
// True once the calling thread's current device has been configured through `mark`.
struct SanPerDevice {
    unsigned long long done[2] = {0ull, 0ull};      // devices 0 .. 127
    bool has(int dev) const { return dev >= 0 && dev < 128 && ((done[dev >> 6] >> (dev & 63)) & 1ull); }
    void mark(int dev) {
        if (dev >= 0 && dev < 128) done[dev >> 6] |= 1ull << (dev & 63);
    }
};
static inline int san_current_device() {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

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
// Total of all 64 lanes, returned wave-uniform.  Pure VALU (DPP butterflies inside each row
// of 16 lanes, row_bcast across rows, readlane 63): no LDS crossbar traffic, unlike __shfl.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float san_dpp_get(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// Kernels that may share the compute units with another stream's MFMA kernel are compiled WITHOUT packed-fp32 instructions.
// Measured on MI355X (scratch/two_stream_probe.py, scratch/attempts/r4_sens_overlap_notes.md): `v_pk_mul_f32 d, a[0:1], v[m1:m2]
// op_sel:[0,1]` (both halves take m2, the HIGH register of the pair) now and then read the LOW register in the low half for the
// last 16 lanes of a wave while a convolution kernel of another stream was resident on the same compute units -- 16 elements of
// a plane off by exactly s * yh * (m1 - m2) in 2 of 3 launches; never when the kernel ran alone, never without the packed form.
#if defined(__HIP_DEVICE_COMPILE__)
#define SAN_NO_PK32 __attribute__((target("no-packed-fp32-ops")))
#else
#define SAN_NO_PK32
#endif

__device__ __forceinline__ float san_wave_total(float v) {
    v += san_dpp_get<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += san_dpp_get<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += san_dpp_get<0x141, 0xf>(v);   // row_half_mirror
    v += san_dpp_get<0x140, 0xf>(v);   // row_mirror: every lane holds its row's sum
    v += san_dpp_get<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v += san_dpp_get<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double san_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// "amax record" of a gradient tensor (fp16-format gradients: the data / weight gradient kernels scale dy by a power of two
// derived from its largest magnitude).  A record is SAN_AMAX_LINES uint32 values, one per 128-byte line (SAN_AMAX_WORDS words
// in all, zeroed by the caller once per step): the kernels that write dy fold each workgroup's maximum into line
// (workgroup id mod 64) with one integer atomic max (float bits of a non-negative value: order-independent, deterministic),
// the readers take the maximum over the 64 lines.  64 separate lines: same-line atomics serialise (~12 ns each; 16 k of them
// on one address cost ~200 us per launch), 64 lines keep that under a microsecond -- and no finalising launch is needed.
#define SAN_AMAX_LINES 64
#define SAN_AMAX_STRIDE 32
#define SAN_AMAX_WORDS (SAN_AMAX_LINES * SAN_AMAX_STRIDE)

// all threads of the workgroup call this at a point every thread reaches; WAVES = waves per workgroup
template <int WAVES>
__device__ __forceinline__ void san_amax_record(unsigned* amax, int wg_linear, float mx) {
    if (!amax) return;                                  // (workgroup-uniform)
    __shared__ float san_amax_red[WAVES];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) san_amax_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, san_amax_red[w]);
        const unsigned bits = __builtin_bit_cast(unsigned, mx);
        if (bits != 0u)
            __hip_atomic_fetch_max(amax + (wg_linear & (SAN_AMAX_LINES - 1)) * SAN_AMAX_STRIDE, bits, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// bits of the tensor's largest magnitude (wave-uniform); every lane of the wave must be active
__device__ __forceinline__ uint32_t san_amax_read(const uint32_t* amax) {
    uint32_t b = amax[(threadIdx.x & 63) * SAN_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)b, o, 64);
        b = b > t ? b : t;
    }
    return b;
}
