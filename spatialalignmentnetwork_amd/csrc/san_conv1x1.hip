// 1x1 convolutions and ConvTranspose2d(2x2, stride 2) as ONE-STAGE GEMMs on the fp16 matrix cores (round 5).
//
// varnet.py:159-192 (TransposeConvBlock), unet.py's 1x1 layers and their data gradients are products
//     D[pixel][cout'] = sum_k A[pixel][k] W[k][cout'],   K = cin <= 576, cout' = cout (1x1) or 4 cout (transposed: + pixel shuffle)
// with 1-2 GFLOP and 10-90 MB per launch at N = 8.  On the tiled convolution kernel (san_conv_bf16.hip, KS = 1) every 24-channel
// chunk of K costs a staging phase and two workgroup barriers for ONE K-step of matrix work: the four transposed convolutions of
// a cascade took 24-48 us each and their data gradients 28-80 us (rocprofv3, in the step), 3.5 ms of a 44.5 ms step, all latency.
// Here a workgroup stages its WHOLE K range of a 64-pixel tile once (lazy affine + LeakyReLU, two fp16 parts, [pixel][channel]
// image in LDS with a conflict-free stride), then runs all K-steps without a barrier: wave w owns the 16 pixels 16 w .. of the
// tile and NG blocks of 16 output channels; activation fragments come from LDS (two 16-byte reads per K-step), weight fragments
// straight from the packed image in L2 / L1 (already in MFMA lane order), requested one K-step ahead.  Same packed weights (two fp16
// parts, 24 channels + 8 zero columns per K-step of 32), same three products per MAC (a1 w1 + a1 w2 + a2 w1), same epilogue
// contract as the tiled kernel: bias, per-tile (count, mean, M2) statistics, pixel shuffle, power-of-two rescale of amax-scaled
// gradient inputs.  Formats: two fp16 parts (the fp32-equivalent mode) and one plain bf16 part (the narrow modes' one-product
// layers: BF1); three bf16 parts and fp8 stay on the tiled kernel.
#include <stdlib.h>

#include "san_common.h"

namespace {

constexpr int kT = 256;
constexpr int kTP = 64;                      // pixels per workgroup (flattened h * w index of one sample)
constexpr int kCKC = 24;                     // input channels per K-step (the packed image's chunk)
constexpr int kMaxChunks = 12;               // K-steps staged at once (288 channels: 78 KB of LDS for both parts)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float fl2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
union Frag {
    u32x4 u;
    h8 h;
    bf8 v;
};
__device__ __forceinline__ uint32_t cvt_pk_b(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));      // round to nearest even
}

__device__ __forceinline__ uint32_t cvt_pk_h(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hf2));      // round to nearest even, denormal results kept
}
// f = a1 + a2 with a1 = fp16(f), a2 = fp16(f - a1) (22 mantissa bits), two values at a time (san_conv_bf16.hip: split2h_pair)
__device__ __forceinline__ void split2h_pair(float f0, float f1, uint32_t& p1, uint32_t& p2) {
    p1 = cvt_pk_h(f0, f1);
    const hf2 h = __builtin_bit_cast(hf2, p1);
    p2 = cvt_pk_h(__builtin_fmaf((float)h[0], -1.f, f0), __builtin_fmaf((float)h[1], -1.f, f1));
}

// LDS bytes per staged pixel and part: room for `chunks` K-steps of 24 channels + the 8 columns the last K-step's fourth lane row
// reads (zero weights, but the operand must be finite); stride = 32 mod 64 bytes: the 16-byte fragment reads of a wave and the
// 16-byte staging writes of 8 consecutive pixels are bank-conflict free (checked exhaustively for every K in use)
__host__ __device__ inline int pixel_stride(int chunks) {
    const int need = chunks * kCKC * 2 + 16;
    return ((need + 31) / 64) * 64 + 32;
}

// NGW: blocks of 16 output channels per WAVE.  A workgroup covers 64 pixels x 4 NGW blocks; wave w owns ALL 64 pixels and the blocks
// cb0 + w, cb0 + w + 4, ..: every weight fragment is fetched by exactly one wave of the workgroup (a pixel-per-wave split had
// all four waves pull the same 12 KB per K-step through their L1 with one step of run-ahead: 25 us per layer, all latency), the
// ring of weight fragments runs kWD K-steps ahead, and a wave's statistics cover the whole tile (no cross-wave merge).
constexpr int kWD = 4;

// BF1: the one-part plain-bf16 form of the narrow-precision modes (san_set_conv_precision(1): one product per MAC on part 0 of a
// bf16-format image, activations rounded to bf16 while they are staged, no amax scale -- bf16 has fp32's exponent range).
template <int NGW, bool SHUFFLE, bool BF1>
__global__ void __launch_bounds__(kT) gemm1x1_f16_kernel(const SanGemm1x1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nn = lane & 15, kg = lane >> 4;
    const int HW = a.H * a.W;
    const int ngrp = a.ngrp, ptiles = a.ptiles;
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;      // each XCD (own L2) gets a contiguous run of the logical order
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int grp = lin % ngrp;                            // channel group fastest: the groups of a tile share its input in L2
    const int pt = (lin / ngrp) % ptiles;
    const int n = lin / (ngrp * ptiles);
    const int p0 = pt * kTP;
    const int cb0 = grp * 4 * NGW + wave;                  // this wave's first block of 16 output channels (then every fourth)

    // gradient input in the fp16 format: x S (S = 2^(13 - floor(log2 max |x|)), exact) rides in the affine, the outputs get 1 / S
    float inS = 1.f, inInvS = 1.f;
    if (!BF1 && a.amax) {
        const uint32_t b = san_amax_read(a.amax);
        int e = (int)((b >> 23) & 255u);
        if (b != 0u) {
            e = e < 14 ? 14 : (e > 250 ? 250 : e);
            inS = __builtin_bit_cast(float, (uint32_t)(267 - e) << 23);
            inInvS = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
        }
    }
    const float lrelu_c = a.in_slope <= 1.f ? __builtin_inff() : -__builtin_inff();      // med3(v, slope v, +-inf) = leaky_relu

    f4 acc[NGW][4];
#pragma unroll
    for (int m = 0; m < NGW; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[m][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int spx = lane;                                  // staging: pixel of the tile (wave w takes the 8-channel groups w, w + 4, ..)
    const bool spx_ok = p0 + spx < HW;
    const float* xpix = a.x + (size_t)(n * a.x_ctot + a.x_coff) * HW + p0 + (spx_ok ? spx : 0);
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp) + lane;

    for (int k0 = 0; k0 < a.chunks; k0 += kMaxChunks) {
        const int kc = min(kMaxChunks, a.chunks - k0);     // K-steps of this pass
        const int S = pixel_stride(kc);
        unsigned char* lds1 = smem;
        unsigned char* lds2 = smem + (size_t)kTP * S;
        // the weight fragments of the first kWD K-steps are requested before the staging phase
        Frag wq[kWD][NGW][2];
        auto load_w = [&](int c, Frag (&dst)[NGW][2]) {
            const u32x4* src = wbase + ((size_t)(k0 + c) * a.nblkp + cb0) * 3 * 64;
#pragma unroll
            for (int m = 0; m < NGW; ++m) {
                dst[m][0].u = src[(m * 4 * 3 + 0) * 64];
                if constexpr (!BF1) dst[m][1].u = src[(m * 4 * 3 + 1) * 64];
            }
        };
#pragma unroll
        for (int d = 0; d < kWD; ++d)
            if (d < kc) load_w(d, wq[d]);
        if (k0) __syncthreads();                            // everyone is done with the previous pass's image
        // ---- stage: (pixel, 8-channel group) units; channels past cin load a clamped (real) channel and meet zero weights
        const int ngroups = kc * 3;
        constexpr int SB = 3;                               // groups per wave in flight: 24 global loads before the first use
        for (int gb = wave; gb < ngroups + 1; gb += 4 * SB) {        // (wave-uniform: the affine entries are scalar loads)
            float v[SB][8];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int c0 = k0 * kCKC + (gb + 4 * b) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[b][i] = (gb + 4 * b < ngroups && spx_ok) ? xpix[(size_t)min(c0 + i, a.cin - 1) * HW] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int g = gb + 4 * b;
                if (g < ngroups + 1) {
                    const int c0 = k0 * kCKC + g * 8;
                    uint32_t q1[4] = {0u, 0u, 0u, 0u}, q2[4] = {0u, 0u, 0u, 0u};
                    if (g < ngroups && spx_ok) {
#pragma unroll
                        for (int i = 0; i < 8; i += 2) {
                            float t0 = v[b][i] * inS, t1 = v[b][i + 1] * inS;
                            if (a.in_scale) {
                                const int ca = n * a.x_ctot + a.x_coff + min(c0 + i, a.cin - 1), cb = n * a.x_ctot + a.x_coff + min(c0 + i + 1, a.cin - 1);
                                t0 = fmaf(v[b][i], a.in_scale[ca] * inS, a.in_shift[ca] * inS);
                                t1 = fmaf(v[b][i + 1], a.in_scale[cb] * inS, a.in_shift[cb] * inS);
                            }
                            t0 = __builtin_amdgcn_fmed3f(t0, t0 * a.in_slope, lrelu_c);
                            t1 = __builtin_amdgcn_fmed3f(t1, t1 * a.in_slope, lrelu_c);
                            if constexpr (BF1) q1[i >> 1] = cvt_pk_b(t0, t1);
                            else split2h_pair(t0, t1, q1[i >> 1], q2[i >> 1]);
                        }
                    }
                    // (group `ngroups` = the zero columns behind the last K-step; pixels past the image are zero rows)
                    *reinterpret_cast<u32x4*>(lds1 + spx * S + g * 16) = u32x4{q1[0], q1[1], q1[2], q1[3]};
                    if constexpr (!BF1) *reinterpret_cast<u32x4*>(lds2 + spx * S + g * 16) = u32x4{q2[0], q2[1], q2[2], q2[3]};
                }
            }
        }
        __syncthreads();
        // ---- K-steps: activation fragments of (pixel 16 b + nn, channels 24 c + 8 kg ..) from LDS, weight fragments from the ring;
        // lane row kg = 3 of a K-step meets zero weights
        const unsigned char* xrow = lds1 + nn * S + kg * 16;
        for (int c = 0; c < kc; c += kWD) {
#pragma unroll
            for (int d = 0; d < kWD; ++d) {
                if (c + d < kc) {
                    Frag x1[4], x2[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        x1[b].u = *reinterpret_cast<const u32x4*>(xrow + (size_t)(16 * b) * S + (c + d) * (kCKC * 2));
                        if constexpr (!BF1) x2[b].u = *reinterpret_cast<const u32x4*>(xrow + (size_t)(16 * b) * S + (size_t)kTP * S + (c + d) * (kCKC * 2));
                    }
#pragma unroll
                    for (int m = 0; m < NGW; ++m)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            if constexpr (BF1) {
                                if constexpr (SHUFFLE) acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[d][m][0].v, x1[b].v, acc[m][b], 0, 0, 0);
                                else acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1[b].v, wq[d][m][0].v, acc[m][b], 0, 0, 0);
                            } else if constexpr (SHUFFLE) {       // D rows = channels: a lane ends with the 4 virtual channels of one real channel
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[d][m][0].h, x1[b].h, acc[m][b], 0, 0, 0);
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[d][m][0].h, x2[b].h, acc[m][b], 0, 0, 0);
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[d][m][1].h, x1[b].h, acc[m][b], 0, 0, 0);
                            } else {                       // D rows = pixels: a lane ends with 4 CONSECUTIVE PIXELS of one channel
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1[b].h, wq[d][m][0].h, acc[m][b], 0, 0, 0);
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x2[b].h, wq[d][m][0].h, acc[m][b], 0, 0, 0);
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1[b].h, wq[d][m][1].h, acc[m][b], 0, 0, 0);
                            }
                        }
                    if (c + d + kWD < kc) load_w(c + d + kWD, wq[d]);       // this slot's next occupant, kWD K-steps ahead
                }
            }
        }
    }

    if (!BF1 && a.w_tail) inInvS *= a.w_tail[1];          // fp16-format weights may be stored x S_w (round 6): exact power of two
    // ------------------------------------------------------------ epilogue
#pragma unroll
    for (int m = 0; m < NGW; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[m][b] = acc[m][b] * inInvS;
    if constexpr (SHUFFLE) {
        // acc[m][b][r] = output channel (cb0 + 4 m) 16 + 4 kg + r at pixel p0 + 16 b + nn
        bool valid[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) valid[b] = p0 + 16 * b + nn < HW;
        if (a.bias) {
#pragma unroll
            for (int m = 0; m < NGW; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (cb0 + 4 * m) * 16 + 4 * kg + r;
                    const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[m][b][r] += bv;
                }
        }
        if (a.part) {
            // (count, mean, M2) of every channel over the tile's 64 pixels: pilot-shifted single pass, DPP row sums over the 16 lanes
            float cnt = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) cnt += valid[b] ? 1.f : 0.f;
            cnt += san_dpp_get<0xB1, 0xf>(cnt);
            cnt += san_dpp_get<0x4E, 0xf>(cnt);
            cnt += san_dpp_get<0x141, 0xf>(cnt);
            cnt += san_dpp_get<0x140, 0xf>(cnt);
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
            for (int m = 0; m < NGW; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pilot = __shfl(acc[m][0][r], lane & 48, 64);    // the tile's first pixel (valid: p0 < HW)
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float e = valid[b] ? acc[m][b][r] - pilot : 0.f;
                        s1 += e;
                        s2 = fmaf(e, e, s2);
                    }
                    s1 += san_dpp_get<0xB1, 0xf>(s1);
                    s2 += san_dpp_get<0xB1, 0xf>(s2);
                    s1 += san_dpp_get<0x4E, 0xf>(s1);
                    s2 += san_dpp_get<0x4E, 0xf>(s2);
                    s1 += san_dpp_get<0x141, 0xf>(s1);
                    s2 += san_dpp_get<0x141, 0xf>(s2);
                    s1 += san_dpp_get<0x140, 0xf>(s1);
                    s2 += san_dpp_get<0x140, 0xf>(s2);
                    const int co = (cb0 + 4 * m) * 16 + 4 * kg + r;
                    if (nn == 0 && co < a.cout) {
                        // the 4 virtual channels of a real channel are 4 interleaved slot sequences of its plane
                        float* dst = a.part + ((size_t)(n * (a.cout >> 2) + (co >> 2)) * (a.slots * 4) + (co & 3)) * 3;
                        float* o = dst + (size_t)pt * 12;
                        o[0] = cnt;
                        o[1] = pilot + s1 * inv;
                        o[2] = fmaxf(s2 - s1 * s1 * inv, 0.f);
                        for (int s = pt + ptiles; s < a.slots; s += ptiles) {   // unused slots: empty records
                            float* z = dst + (size_t)s * 12;
                            z[0] = z[1] = z[2] = 0.f;
                        }
                    }
                }
        }
        // virtual channel 4 c + 2 dy + dx of input pixel (y, x) is output pixel (2 y + dy, 2 x + dx) of channel c; a lane holds
        // r = 0..3 = the four positions of one real channel: two 8-byte stores
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (valid[b]) {
                const int q = p0 + 16 * b + nn;
                const int yy = q / a.W, xx = q - yy * a.W;
#pragma unroll
                for (int m = 0; m < NGW; ++m) {
                    const int co = (cb0 + 4 * m) * 16 + 4 * kg;
                    if (co < a.cout) {
                        float* dst = a.y + (size_t)(n * a.y_ctot + a.y_coff + (co >> 2)) * (4 * (size_t)HW) + (size_t)(2 * yy) * (2 * a.W) + 2 * xx;
                        *reinterpret_cast<fl2*>(dst) = fl2{acc[m][b][0], acc[m][b][1]};
                        *reinterpret_cast<fl2*>(dst + 2 * a.W) = fl2{acc[m][b][2], acc[m][b][3]};
                    }
                }
            }
        }
    } else {
        // acc[m][b][r] = output channel (cb0 + 4 m) 16 + nn at pixel p0 + 16 b + 4 kg + r
        bool v1[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[b][r] = p0 + 16 * b + 4 * kg + r < HW;
        if (a.bias) {
#pragma unroll
            for (int m = 0; m < NGW; ++m) {
                const int co = (cb0 + 4 * m) * 16 + nn;
                const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[m][b] += f4{bv, bv, bv, bv};
            }
        }
        if (a.part) {
            float cnt = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) cnt += v1[b][r] ? 1.f : 0.f;
            cnt += __shfl_xor(cnt, 16, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
            for (int m = 0; m < NGW; ++m) {
                const float pilot = __shfl(acc[m][0][0], nn, 64);            // the tile's first pixel of this channel
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = v1[b][r] ? acc[m][b][r] - pilot : 0.f;
                        s1 += e;
                        s2 = fmaf(e, e, s2);
                    }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                const int co = (cb0 + 4 * m) * 16 + nn;
                if (kg == 0 && co < a.cout) {
                    float* dst = a.part + (size_t)(n * a.cout + co) * a.slots * 3;
                    float* o = dst + (size_t)pt * 3;
                    o[0] = cnt;
                    o[1] = pilot + s1 * inv;
                    o[2] = fmaxf(s2 - s1 * s1 * inv, 0.f);
                    for (int s = pt + ptiles; s < a.slots; s += ptiles) {       // unused slots: empty records
                        float* z = dst + (size_t)s * 3;
                        z[0] = z[1] = z[2] = 0.f;
                    }
                }
            }
        }
        const bool aligned = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
#pragma unroll
        for (int m = 0; m < NGW; ++m) {
            const int co = (cb0 + 4 * m) * 16 + nn;
            if (co < a.cout) {
                float* dst = a.y + (size_t)(n * a.y_ctot + a.y_coff + co) * HW + p0 + 4 * kg;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (aligned && v1[b][3]) {
                        *reinterpret_cast<f4*>(dst + 16 * b) = acc[m][b];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (v1[b][r]) dst[16 * b + r] = acc[m][b][r];
                    }
                }
            }
        }
    }
}

// SAN_CONV1X1_GEMM=0 in the environment: off from the start (same-box A/B of whole steps)
int g_gemm1x1 = (getenv("SAN_CONV1X1_GEMM") && atoi(getenv("SAN_CONV1X1_GEMM")) == 0) ? 0 : 1;

template <int NGW, bool BF1>
int launch_ng(const SanGemm1x1Args& a, size_t lds, hipStream_t s) {
    const dim3 grid(a.ptiles * a.ngrp * a.N);
    static SanPerDevice configured[2];
    const int k = a.shuffle ? 1 : 0;
    const int dev__ = san_current_device();
    if (!configured[k].has(dev__)) {
        const void* fn = a.shuffle ? reinterpret_cast<const void*>(&gemm1x1_f16_kernel<NGW, true, BF1>) : reinterpret_cast<const void*>(&gemm1x1_f16_kernel<NGW, false, BF1>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
            san_set_error("cannot reserve 96 KB of LDS for the 1x1 GEMM");
            return SAN_E_UNSUPPORTED;
        }
        configured[k].mark(dev__);
    }
    if (a.shuffle) hipLaunchKernelGGL((gemm1x1_f16_kernel<NGW, true, BF1>), grid, dim3(kT), lds, s, a);
    else hipLaunchKernelGGL((gemm1x1_f16_kernel<NGW, false, BF1>), grid, dim3(kT), lds, s, a);
    return SAN_OK;
}

}  // namespace

// (internal: called by conv_bf16x3_run in san_conv_bf16.hip for KS = 1 layers on fp16-format weights)
bool san_gemm1x1_enabled() { return g_gemm1x1 != 0; }

int san_gemm1x1_f16_run(SanGemm1x1Args a, void* stream) {
    const int nblk = san_cdiv(a.cout, 16);
    a.ptiles = san_cdiv(a.H * a.W, kTP);
    // blocks per wave: the largest of 3, 2, 1 that still gives the chip ~1.5 workgroups per compute unit, then the smallest with the
    // same number of channel groups (fewer blocks that do not exist)
    int NGW = 1;
    for (int ngw = 3; ngw >= 1; --ngw)
        if (ngw == 1 || (long long)a.ptiles * a.N * san_cdiv(nblk, 4 * ngw) >= 384) {
            NGW = ngw;
            break;
        }
    while (NGW > 1 && san_cdiv(nblk, 4 * (NGW - 1)) == san_cdiv(nblk, 4 * NGW)) --NGW;
    while (NGW > 1 && san_cdiv(nblk, 4 * NGW) * 4 * NGW - nblk > 4) --NGW;      // (the packed image is padded by 4 blocks)
    a.ngrp = san_cdiv(nblk, 4 * NGW);
    if (a.ngrp * 4 * NGW > a.nblkp) {
        san_set_error("1x1 GEMM: %d channel blocks in the packed image, %d wanted", a.nblkp, a.ngrp * 4 * NGW);
        return SAN_E_ARG;
    }
    if (a.part && a.slots < a.ptiles) {
        san_set_error("1x1 GEMM: %d statistics slots for %d pixel tiles", a.slots, a.ptiles);
        return SAN_E_ARG;
    }
    const int kc = a.chunks < kMaxChunks ? a.chunks : kMaxChunks;
    const size_t lds = (size_t)(a.bf1 ? 1 : 2) * kTP * pixel_stride(kc);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (a.bf1) {
        switch (NGW) {
            case 1: rc = launch_ng<1, true>(a, lds, s); break;
            case 2: rc = launch_ng<2, true>(a, lds, s); break;
            default: rc = launch_ng<3, true>(a, lds, s); break;
        }
    } else {
        switch (NGW) {
            case 1: rc = launch_ng<1, false>(a, lds, s); break;
            case 2: rc = launch_ng<2, false>(a, lds, s); break;
            default: rc = launch_ng<3, false>(a, lds, s); break;
        }
    }
    if (rc != SAN_OK) return rc;
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

extern "C" int san_conv1x1_gemm_enable(int on) {
    const int prev = g_gemm1x1;
    if (on >= 0) g_gemm1x1 = on ? 1 : 0;
    return prev;
}
